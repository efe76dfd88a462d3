"""pointnav-vo_amd — MI355X-native drop-in for PointNav-VO's visual-odometry hot path.

Sub-modules:
  model_spec   architecture/state_dict description (no torch)
  synth        deterministic synthetic weights / inputs (numpy only)
  _lib         ctypes binding of csrc/libpnvo.so (the C ABI of include/pnvo.h); fails loudly if missing
  registry     mirror of the reference's baseline_registry VO-model API
  vo_cnn       nn.Module mirrors registered under the reference's model names (HIP forward)
  trainer      mirror of BaseRLTrainerWithVO (_setup_vo_model / _compute_local_delta_states_from_vo)
"""
__version__ = "0.1.0"
