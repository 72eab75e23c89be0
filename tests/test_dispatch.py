"""The dispatcher, pinned: (env kind, qubits, batch, options, list length) -> state layout and kernel family, for every row of DESIGN.md's
kernel table.  `qg_plan_query` (include/qgym.h) answers from the decision functions the launch paths themselves call
(qiskit_gym_amd/csrc/qgym_plan.hpp) and needs no GPU: a threshold edited there turns this CPU test red instead of silently routing a
configuration to a slower-but-correct kernel."""
import ctypes as C

import pytest

from qiskit_gym_amd import _lib

LAYOUT, STEP, FUSED, RESET_DONE, OBS_DENSE, OBS_PACKED, STATE_I64, TRACK_DENSE, RESET_DONE_STEP = range(9)
B = 65536


def plan(kind, n, op, batch=B, arg=0, nonsymplectic=0, num_actions=170, **cfg):
    L = _lib.load()
    c = _lib.make_config(kind, n, **cfg)
    buf = C.create_string_buffer(96)
    rc = L.qg_plan_query(C.byref(c), batch, num_actions, op, arg, nonsymplectic, buf, len(buf))
    return buf.value.decode() if rc == 0 else rc


PLAIN = dict(add_inverts=False, add_perms=False, track_solution=False)
DEFAULT = dict(add_inverts=True, add_perms=False, track_solution=True)

LAYOUTS = [
    # kind, qubits, options -> layout                                    (qgym_plan.hpp handle_plan)
    ("permutation", 9, PLAIN, "PERM"), ("permutation", 16, PLAIN, "PERM"), ("permutation", 17, PLAIN, "PERMB"), ("permutation", 256, PLAIN, "PERMB"),
    ("linear_function", 3, PLAIN, "LF8"), ("linear_function", 8, PLAIN, "LF8"), ("linear_function", 8, DEFAULT, "LF8"),
    ("linear_function", 9, PLAIN, "TILE"), ("linear_function", 32, PLAIN, "TILE"), ("linear_function", 33, PLAIN, "TILE64"), ("linear_function", 64, PLAIN, "TILE64"),
    ("linear_function", 9, DEFAULT, "LFD"), ("linear_function", 64, DEFAULT, "LFD"),
    ("clifford", 3, PLAIN, "TILE"), ("clifford", 16, PLAIN, "TILE"), ("clifford", 16, DEFAULT, "TILE"), ("clifford", 17, PLAIN, "TILE64"), ("clifford", 32, DEFAULT, "TILE64"),
    ("pauli", 20, dict(max_rotations=5), "PTILE-compact"), ("pauli", 24, dict(max_rotations=5), "PTILE-compact"),
    ("pauli", 25, dict(max_rotations=5), "PTILE"), ("pauli", 20, dict(max_rotations=9), "PTILE"), ("pauli", 20, dict(max_rotations=5, final_pauli_layers=9), "PTILE"),
]


@pytest.mark.parametrize("kind,n,cfg,want", LAYOUTS)
def test_layout(kind, n, cfg, want):
    assert plan(kind, n, LAYOUT, **cfg) == want


def test_limits_are_reported_as_unsupported():
    for kind, n, cfg in (("permutation", 257, PLAIN), ("linear_function", 65, PLAIN), ("clifford", 33, PLAIN), ("pauli", 33, {}),
                         ("pauli", 20, dict(max_rotations=33)), ("pauli", 20, dict(max_rotations=5, final_pauli_layers=40))):
        assert plan(kind, n, LAYOUT, **cfg) == -3, (kind, n)


STEPS = [
    # kind, qubits, options, nonsymplectic -> env.step() kernel, fused 128-step rollout kernel
    ("clifford", 16, PLAIN, 0, "qm_step1_kernel", "qm_fused_lds_kernel"),                      # BASELINE config 3: the headline
    ("clifford", 16, dict(PLAIN, track_solution=True), 0, "qm_step1_kernel", "qm_step_kernel"),
    ("clifford", 16, DEFAULT, 0, "qm_inv2_kernel", "qm_step_kernel<inv>"),                     # the reference's default options
    ("clifford", 16, DEFAULT, 1, "qm_step_kernel<gauss-jordan>", "qm_step_kernel<gauss-jordan>"),
    ("clifford", 5, DEFAULT, 0, "qm_inv2_kernel", "qm_step_kernel<inv>"),
    ("clifford", 24, PLAIN, 0, "q64_step1_kernel", "q64_fused_lds_kernel"),
    ("clifford", 24, DEFAULT, 0, "q64_inv2_kernel", "q64_step_kernel<inv>"),
    ("clifford", 24, DEFAULT, 1, "q64_step_kernel<gauss-jordan>", "q64_step_kernel<gauss-jordan>"),
    ("linear_function", 8, PLAIN, 0, "word_step_kernel", "word_step_kernel"),                   # config 2
    ("linear_function", 24, PLAIN, 0, "qm_step1_kernel", "qm_fused_lds_kernel"),
    ("linear_function", 24, DEFAULT, 0, "lfd_step_kernel", "lfd_step_kernel"),
    ("linear_function", 48, PLAIN, 0, "q64_step1_kernel", "q64_fused_lds_kernel"),
    ("permutation", 9, PLAIN, 0, "word_step_kernel", "word_step_kernel"),                       # config 1
    ("permutation", 27, PLAIN, 0, "permb_step1_kernel", "permb_step_kernel"),
    ("permutation", 27, DEFAULT, 0, "permb_step_kernel", "permb_step_kernel"),
    ("pauli", 20, dict(max_rotations=5, add_perms=False, track_solution=False), 0, "ptile_step1c_kernel", "ptile_fused1c_kernel"),  # config 5
    ("pauli", 20, dict(max_rotations=5, add_perms=False, track_solution=True), 0, "ptile_step1c_kernel", "ptile_step_kernel"),
    ("pauli", 20, dict(max_rotations=5, add_perms=True, track_solution=False), 0, "ptile_step1c_kernel", "ptile_step_kernel"),
    ("pauli", 28, dict(max_rotations=5, add_perms=False, track_solution=False), 0, "ptile_step1_kernel", "ptile_step_kernel"),
]


@pytest.mark.parametrize("kind,n,cfg,nonsymp,step,fused", STEPS)
def test_step_kernels(kind, n, cfg, nonsymp, step, fused):
    assert plan(kind, n, STEP, nonsymplectic=nonsymp, **cfg) == step
    assert plan(kind, n, FUSED, arg=128, nonsymplectic=nonsymp, **cfg) == fused


def test_layer_weights_take_the_feature_kernels_and_an_empty_gateset_the_register_kernel():
    layered = dict(PLAIN, w_n_layers=0.1)
    assert plan("clifford", 16, STEP, **layered) == "qm_step1_kernel" and plan("clifford", 16, FUSED, arg=64, **layered) == "qm_step_kernel"
    assert plan("clifford", 16, FUSED, arg=64, num_actions=0, **PLAIN) == "qm_step_kernel"
    assert plan("clifford", 16, FUSED, arg=1, **PLAIN) == "qm_step1_kernel"  # a "fused" rollout of one step is env.step()


RESETS = [
    # kind, qubits, batch, difficulty, finished envs -> which scramble resets them    (list_reset_path: tree up to 4 096 envs of >= 64
    ("clifford", 16, B, 256, 512, "scramble_tree"),      #  draws; 16 lanes per env: count * 32 <= B; else one thread per env)
    ("clifford", 16, B, 256, 4096, "scramble_tree"),
    ("clifford", 16, B, 256, 4097, "scramble_flat"),
    ("clifford", 16, 4 * B, 256, 4096, "scramble_tree"),
    ("clifford", 16, 4 * B, 256, 4097, "scramble_coop"),
    ("clifford", 16, 4 * B, 256, 8192, "scramble_coop"),
    ("clifford", 16, 4 * B, 256, 8193, "scramble_flat"),
    ("clifford", 16, B, 64, 512, "scramble_tree"),
    ("clifford", 16, B, 63, 512, "scramble_coop"),
    ("clifford", 16, 1024, 256, 1024, "scramble_tree"),   # (walked in rounds by B / 8 workgroups)
    ("clifford", 16, 63, 256, 1, "scramble_flat"),        # batches below 64 envs never take the cooperative paths
    ("linear_function", 12, B, 256, 512, "scramble_tree"),
    ("clifford", 24, B, 256, 512, "scramble_tree64"),
    ("clifford", 24, B, 256, 1500, "scramble_tree64"),
    ("clifford", 24, 4 * B, 256, 6000, "scramble_coop"),
    ("clifford", 24, B, 32, 4000, "scramble_flat"),
    ("pauli", 20, B, 128, 512, "compact_done + ptile_reset_tree_kernel"),
    ("pauli", 20, B, 128, B // 32, "compact_done + ptile_reset_tree_kernel"),
    ("pauli", 20, B, 128, B // 32 + 1, "compact_done + ptile_generate_kernel"),
    ("pauli", 20, B, 32, 512, "compact_done + ptile_generate_kernel"),
    ("pauli", 20, 4095, 128, 40, "ptile_generate_kernel"),
    ("linear_function", 8, B, 64, 512, "word_init_kernel"),   # (16 lanes per finished env, decided per wave)
    ("permutation", 9, B, 16, 512, "word_init_kernel"),
    ("permutation", 27, B, 64, 512, "init_kernel"),
]


@pytest.mark.parametrize("kind,n,batch,difficulty,count,want", RESETS)
def test_reset_done_paths(kind, n, batch, difficulty, count, want):
    cfg = dict(difficulty=difficulty, add_perms=False, track_solution=False)
    if kind != "pauli":
        cfg["add_inverts"] = False
    assert plan(kind, n, RESET_DONE, batch=batch, arg=count, **cfg) == want


def test_reset_done_step_in_one_launch():
    """qg_vec_reset_done_step: which handles have a kernel that resets the finished envs and steps every env in ONE launch -- and take it: where it is the faster
    form (long scrambles: the resets are trees; qgym_plan.hpp reset_step_pays)."""
    LONG = dict(difficulty=256)  # (episodes of 128 steps: clifford.rs:317 with the default depth_slope 2 and max_depth 128)
    assert plan("clifford", 16, RESET_DONE_STEP, **PLAIN, **LONG).startswith("qm_reset_step_kernel")
    assert plan("linear_function", 24, RESET_DONE_STEP, **PLAIN, **LONG).startswith("qm_reset_step_kernel")
    assert plan("clifford", 24, RESET_DONE_STEP, **PLAIN, **LONG).startswith("q64_reset_step_kernel")            # 64-bit rows
    assert plan("linear_function", 40, RESET_DONE_STEP, **PLAIN, **LONG).startswith("q64_reset_step_kernel")
    assert plan("clifford", 16, RESET_DONE_STEP, **DEFAULT, **LONG).startswith("qm_reset_inv2_step_kernel")   # the reference's default options: two lanes per env
    assert plan("clifford", 7, RESET_DONE_STEP, **DEFAULT, **LONG).startswith("qm_reset_inv2_step_kernel")
    assert plan("clifford", 16, RESET_DONE_STEP, nonsymplectic=1, **DEFAULT, **LONG) == "two launches"          # (some env needs the Gauss-Jordan inversion)
    # short scrambles / short episodes: the two launches are as fast or faster
    assert plan("clifford", 16, RESET_DONE_STEP, difficulty=8, **DEFAULT) == "two launches"                     # (not trees: the lane-per-env resets read back)
    assert plan("clifford", 16, RESET_DONE_STEP, difficulty=1, **PLAIN) == "two launches"                       # (episodes of two steps: half of the batch per step)
    assert plan("clifford", 16, RESET_DONE_STEP, difficulty=8, **PLAIN).startswith("qm_reset_step_kernel")
    assert plan("clifford", 24, RESET_DONE_STEP, difficulty=16, **PLAIN) == "two launches"                      # (64-bit rows behind the 16-lane resets)
    assert plan("clifford", 24, RESET_DONE_STEP, difficulty=4, **PLAIN).startswith("q64_reset_step_kernel")
    assert plan("linear_function", 8, RESET_DONE_STEP, num_actions=28, **PLAIN) == "word_reset_step_kernel"      # config 2's env: always
    assert plan("linear_function", 8, RESET_DONE_STEP, num_actions=28, **DEFAULT) == "word_reset_step_kernel"    # ... with the reference's defaults
    assert plan("permutation", 9, RESET_DONE_STEP, num_actions=12, **DEFAULT) == "word_reset_step_kernel"
    assert plan("linear_function", 8, RESET_DONE_STEP, num_actions=0, **PLAIN) == "two launches"                  # (an empty gateset: reset is an error)
    for kind, n, cfg in (("clifford", 24, DEFAULT), ("linear_function", 24, DEFAULT), ("permutation", 27, PLAIN), ("pauli", 20, {})):
        assert plan(kind, n, RESET_DONE_STEP, **cfg, **LONG) == "two launches", (kind, n)


def test_observation_and_state_paths():
    assert plan("clifford", 16, OBS_DENSE, **PLAIN) == "qm_dense_stream_kernel"
    assert plan("clifford", 8, OBS_DENSE, **PLAIN) == "qm_dense_stream_kernel"
    assert plan("clifford", 5, OBS_DENSE, **PLAIN) == "qm_dense_stream_any_kernel"   # 10 rows: 16-byte chunks cross rows and envs
    assert plan("clifford", 12, OBS_DENSE, **PLAIN) == "qm_dense_stream_any_kernel"
    assert plan("linear_function", 9, OBS_DENSE, **PLAIN) == "qm_dense_stream_any_kernel"
    assert plan("linear_function", 32, OBS_DENSE, **PLAIN) == "qm_dense_stream_kernel"
    assert plan("linear_function", 16, OBS_DENSE, **PLAIN) == "qm_dense_stream_kernel"
    assert plan("clifford", 24, OBS_DENSE, **PLAIN) == "row words + expand"
    assert plan("linear_function", 24, OBS_DENSE, **DEFAULT) == "row words + expand"
    assert plan("pauli", 20, OBS_DENSE, max_rotations=5) == "row words + expand"
    assert plan("clifford", 16, OBS_PACKED, **PLAIN) == "qm_pack_kernel"
    assert plan("clifford", 24, OBS_PACKED, **PLAIN) == "export_kernel"
    for kind, n, cfg in (("clifford", 16, PLAIN), ("clifford", 24, PLAIN), ("linear_function", 24, DEFAULT), ("pauli", 20, {})):
        assert plan(kind, n, STATE_I64, batch=64, **cfg) == "row words / bit stream + streaming kernel"
        assert plan(kind, n, STATE_I64, batch=63, **cfg) == "init / export kernel"  # (the scalar qg_env_* handles are batches of one)
    assert plan("linear_function", 8, STATE_I64, **PLAIN) == "init / export kernel"


def test_track_dense_modes():
    assert plan("clifford", 16, TRACK_DENSE, **PLAIN) == "in-step"
    assert plan("clifford", 16, TRACK_DENSE, **DEFAULT) == "in-step"   # add_inverts: the two-lanes-per-env step rewrites the env when its coin fires
    assert plan("clifford", 8, TRACK_DENSE, **DEFAULT) == "refresh"    # ... for 16 qubits only: other sizes get a full rewrite after the step
    assert plan("clifford", 8, TRACK_DENSE, **PLAIN) == "in-step"
    assert plan("linear_function", 32, TRACK_DENSE, **PLAIN) == "in-step"
    for kind, n in (("clifford", 5), ("clifford", 20), ("linear_function", 8), ("permutation", 9), ("pauli", 20)):
        assert plan(kind, n, TRACK_DENSE, **({} if kind == "pauli" else PLAIN)) == -3
