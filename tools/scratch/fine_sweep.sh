for T in 224 520 1100; do echo "== PNVO_FINE_BELOW=$T"; PNVO_FINE_BELOW=$T python tools/bench_batch_sweep.py 2>/dev/null | grep -E "B= +(48|64|96|128|192|256) "; done
