"""bench.py's command line and bookkeeping, without a GPU: the driver starts it with `--gpus N --steps K --warmup W` and with no
flags at all; a NameError at import or parse time would cost the round its measurement."""
import importlib
import os
import sys


def _bench(argv):
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        bench = importlib.import_module("bench")
        return bench, bench.parse()
    finally:
        sys.argv = old


def test_default_flags():
    bench, args = _bench([])
    assert args.gpus == 1 and args.objects == 8 and args.frames_per_step == 32
    assert args.steps >= 1 and args.warmup >= 0


def test_driver_flags():
    _, args = _bench(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert (args.gpus, args.steps, args.warmup) == (8, 20, 5)


def test_winograd_accounting_follows_the_dispatch_threshold():
    bench, _ = _bench([])
    # 256 crops: every 128-channel 3x3 at >= 16x16 and both 64-channel ones run in Winograd form (csrc/conv_wino.hip: >= 256 tiles)
    full = bench.winograd_saved_gflop_per_crop(256)
    per = lambda hw, ch: 2.0 * hw * hw * ch * ch * 9 * (1 - 16 / 36) / 1e9
    assert abs(full - (9 * per(64, 128) + 12 * per(32, 128) + 12 * per(16, 128) + per(128, 64) + per(64, 64))) < 1e-9
    # 8 crops (one frame per call): 64x64 has 8*8*4 = 256 tiles and 128x128 has 1024; 32x32 (64 tiles) and 16x16 do not qualify
    one = bench.winograd_saved_gflop_per_crop(8)
    assert abs(one - (9 * per(64, 128) + per(128, 64) + per(64, 64))) < 1e-9
    # the fp16 pipe (default) takes a call from ONE crop up (32 tiles at 64x64: csrc/net.hip); the other pipes from 256 tiles
    assert bench.winograd_saved_gflop_per_crop(1) == one
    os.environ["SUO_F16X2"] = "0"
    try:
        assert bench.winograd_saved_gflop_per_crop(1) < bench.winograd_saved_gflop_per_crop(8) == one
    finally:
        del os.environ["SUO_F16X2"]
    assert bench.GFLOP_PER_CROP - bench.GFLOP_SKIPPED_PER_CROP - full > 0


def test_committed_pmc_summary_feeds_roofline_traffic():
    """roofline.traffic of the default run (256 crops per launch) comes from profiles/pmc_dominant_conv.json: the committed summary
    must be for that kernel and launch shape, and sane against the algorithmic bytes."""
    bench, args = _bench([])
    L = args.objects * args.frames_per_step
    t = bench.dominant_kernel_traffic(L)
    assert t is not None
    algorithmic = 4.0 * L * 64 * 64 * (128 + 256 + 256)
    assert 0.9 * algorithmic < t < 1.5 * algorithmic
    assert bench.dominant_kernel_traffic(L + 8) is None


def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must become two ranks (the driver's BENCH command has that shape):
    the parent starts torch.distributed.run as a child before touching the GPU and relays rank 0's line.  --dry-run stops after
    the rendezvous; gloo stands in for RCCL on this GPU-less box."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SUO_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(x) for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, out.stdout
    assert lines[0]["n_ranks_seen"] == 2 and lines[0]["n_gpus"] == 2 and lines[0]["rccl_backend"] == "gloo"


def test_one_gpu_stays_one_process():
    bench, args = _bench(["--gpus", "1", "--dry-run"])
    assert args.dry_run and args.gpus == 1


def test_line_bookkeeping_of_round_6():
    """The legs live in bench_legs/ and bench.py re-exports what the tools import; `dtype` names the arithmetic; the whole-call roofline is bytes over time
    against a copy's rate; the committed PMC summary is the one of the kernel the line names (its template signature gained a parameter this round)."""
    bench, args = _bench(["--no-tless-leg"])
    assert args.no_tless_leg
    from bench_legs import common, path, roofline
    assert bench.conv_roofline is roofline.conv_roofline and bench.slam_leg is path.slam_leg and bench.tless_leg is path.tless_leg
    assert set(common.DTYPE) == {"f16x2", "bf16x3", "f32"} and common.DTYPE[bench.matrix_pipe()].startswith("f32")
    w = roofline.whole_call({"total": 63e9, "launches": 84, "gemm_1x1": 23e9, "conv3x3_and_fused_tails": 40e9}, 10.0, 4)
    assert w["bound"] == "hbm" and abs(w["achieved"] - 6300.0) < 1e-6 and abs(w["frac_of_achievable_6300"] - 1.0) < 1e-9 and abs(w["frac"] - 6300 / 8000) < 1e-4
    assert set(w["by_kind"]) == {"gemm_1x1", "conv3x3_and_fused_tails"} and w["launches_per_call"] == 84
    import json
    rec = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_dominant_conv.json")))
    assert rec["kernel"].replace(" ", "").startswith(bench.dominant_kernel_name())
