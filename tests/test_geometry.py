"""CPU: property tests of the goal / pose bookkeeping (parity unpinned: the reference delegates to absent packages)."""
import numpy as np

from pointnav_vo_amd import geometry as g


def test_goal_update_identity_and_translation():
    goal = np.array([1.0, 0.0, -2.0])
    out = g.compute_goal_pos(goal, (0.0, 0.0, 0.0))
    np.testing.assert_allclose(out["cartesian"], goal)
    np.testing.assert_allclose(out["polar"], [np.hypot(2.0, 1.0), -np.arctan2(1.0, 2.0)], rtol=1e-6)
    # moving forward by 0.25 (habitat: -z is forward) brings a goal straight ahead 0.25 closer
    out = g.compute_goal_pos(np.array([0.0, 0.0, -2.0]), (0.0, -0.25, 0.0))
    np.testing.assert_allclose(out["polar"], [1.75, 0.0], atol=1e-6)


def test_goal_update_rotation_about_y():
    # turning left by 90 deg (dyaw = +pi/2) moves a goal straight ahead to the agent's right: phi flips sign convention
    out = g.compute_goal_pos(np.array([0.0, 0.0, -1.0]), (0.0, 0.0, np.pi / 2))
    np.testing.assert_allclose(out["cartesian"], [1.0, 0.0, 0.0], atol=1e-12)
    np.testing.assert_allclose(out["polar"], [1.0, -np.pi / 2], atol=1e-6)
    assert abs(out["polar"][0] - 1.0) < 1e-6     # rotations preserve the distance


def test_global_state_composes_with_goal_update():
    """Tracking the agent globally and re-expressing a fixed world goal == updating the goal locally step by step."""
    rng = np.random.default_rng(0)
    rot, pos = np.array([0.0, 0.0, 0.0, 1.0]), np.zeros(3)
    goal_world = np.array([2.0, 0.0, -3.0])
    goal_local = goal_world.copy()
    for _ in range(20):
        d = (rng.uniform(-0.1, 0.1), rng.uniform(-0.3, 0.0), rng.uniform(-0.3, 0.3))
        rot, pos = g.compute_global_state((rot, pos), d)
        goal_local = g.compute_goal_pos(goal_local, d)["cartesian"]
        rinv = np.array([-rot[0], -rot[1], -rot[2], rot[3]])
        np.testing.assert_allclose(goal_local, g._quat_rotate(rinv, goal_world - pos), atol=1e-10)
    assert abs(np.linalg.norm(rot) - 1.0) < 1e-12


def test_batched_goal_update_is_the_per_environment_one():
    rng = np.random.default_rng(3)
    goals = rng.normal(size=(17, 3)) * 4.0
    deltas = np.stack([rng.normal(size=17) * 0.2, rng.normal(size=17) * 0.3, rng.uniform(-np.pi, np.pi, size=17)], axis=1)
    deltas[0] = 0.0
    out = g.compute_goal_pos_batch(goals, deltas)
    for e in range(17):
        one = g.compute_goal_pos(goals[e], deltas[e])
        assert np.allclose(out["cartesian"][e], one["cartesian"], rtol=0, atol=1e-15)
        assert np.array_equal(out["polar"][e], one["polar"]) or np.allclose(out["polar"][e], one["polar"], rtol=0, atol=1e-7)
