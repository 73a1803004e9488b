"""bench.py end to end on the GPU: the JSON contract of the default workload and the two other
BASELINE workloads (short runs)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"}


def _run(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup",
                        "1", *flags], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_bench_default_line_has_roofline_and_cpu_baseline():
    d = _run()
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "scenes/s" and d["value"] > 0 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert abs(d["value"] - 8 * 1000.0 / d["ms_per_step"]) <= 0.02 * d["value"]
    # `value` is measured on the loop that FEEDS data (--rotate 4: distinct batches from pinned host
    # memory, the copy inside the timed region); the resident-batch loop of rounds 1-4 rides along
    assert d["rotate"] == 4 and d["h2d_bytes_per_step"] > 8 * 40000 * 16
    assert abs(d["value_resident"] - 8 * 1000.0 / d["ms_per_step_resident"]) <= 0.02 * d["value_resident"]
    assert 0.7 * d["value_resident"] < d["value"] < 1.3 * d["value_resident"]
    assert 0 < d["host_ms_per_step"] < 20 and "bf16" in d["products"]
    assert d["config"]["hip_graphs"] is True and "workload" in d["config"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["algorithmic_bytes"] == 38516736
    # traffic comes from the committed PMC passes (labelled so): the fused kernel moves LESS than
    # the pair's algorithmic bytes -- it never re-reads the index array and reads 16-byte records
    assert roof["traffic"] is None or (roof["traffic_source"] and 0 < roof["traffic"] < 2 * roof["algorithmic_bytes"])
    forms = roof["forms"]
    assert set(forms) == {"layer", "layer_no_plan", "self_contained", "reference_api_3_calls",
                          "reference_api_3_calls_cached_lists", "layer_rotating_inputs",
                          "layer_cloud_R", "layer_cloud_step"}
    # the sampling kernel's query plans pay (round 6); the step's own launch of the kernel rides
    # along, from the committed rocprofv3 summary of the timed steps
    assert forms["layer"]["us"] < 1.05 * forms["layer_no_plan"]["us"]
    assert roof["in_step_us"] is None or (roof["in_step_source"] and abs(
        roof["in_step_frac"] - roof["algorithmic_bytes"] / (roof["in_step_us"] * 1e-6) / 8e12) < 1e-3)
    # (replays on one input set find their reads in L2 / MALL; twelve rotating sets do not)
    assert 0.8 * forms["layer"]["us"] < forms["layer_rotating_inputs"]["us"] < 2.5 * forms["layer"]["us"]
    # the headline duration is the SLOWEST of the three clouds the kernel is quoted on
    per_cloud = [forms[k]["us"] for k in ("layer", "layer_cloud_R", "layer_cloud_step")]
    assert abs(max(per_cloud) - roof["duration_us"]) < 1e-6
    # density independence of the query kernel: the timed step's own batch (dense object
    # clusters) within 1.6x of the uniform cloud (round 3: 86 us against 18 us)
    assert forms["layer_cloud_step"]["us"] < 1.6 * forms["layer"]["us"]
    assert forms["layer"]["us"] < forms["self_contained"]["us"] < forms["reference_api_3_calls"]["us"]
    # with the cloud's cell lists found in _ext's cache the three reference calls skip the build
    assert forms["layer"]["us"] < forms["reference_api_3_calls_cached_lists"]["us"] < \
        forms["reference_api_3_calls"]["us"]
    assert d["time_op_eager_fallbacks"] == []
    # BASELINE configs[2] and configs[3] ride in the same line (short passes outside the timed region)
    for name, scenes in (("sunrgbd", 16), ("semi", 12)):
        w = d["workloads"][name]
        assert w["per_gpu_batch"] == scenes and w["hip_graphs"] is True and w["value"] > 0
        assert abs(w["value"] - scenes * 1000.0 / w["ms_per_step"]) <= 0.02 * w["value"]
        # the loop that feeds data, as the headline; host time and bytes per step ride along
        assert 0 < w["host_ms_per_step"] < 20 and w["h2d_bytes_per_step"] > scenes * 20000 * 12
        assert 0.7 * w["value_resident"] < w["value"] < 1.3 * w["value_resident"]
    assert d["ms_per_step_no_prefetch"] > d["ms_per_step"]
    kernels = d["kernels"]
    for name in ("fps_40000_2048", "ball_query_sa1", "group_xyz_sa1", "group_feat_sa1",
                 "query_and_group_sa1_fused_kernel", "three_nn_gridconv", "three_interpolate_gridconv",
                 "iou3d_2048x512", "mlp_fwd_sa2_128x128", "mlp_gram_bwd_sa1_128x64",
                 "mlp_fwd_sa2_256x128_no_store", "mlp_gram_bwd_sa2_256x128"):
        assert kernels[name]["us"] > 0 and "bound" in kernels[name], name
    assert 0 < kernels["mlp_fwd_sa2_128x128"]["frac"] < 1
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and 0 < cpu["value"] < d["value"]


@pytest.mark.gpu
@pytest.mark.parametrize("workload,scenes", [("semi", 12), ("sunrgbd", 16)])
def test_bench_other_workloads(workload, scenes):
    d = _run("--workload", workload, "--no-kernels", "--no-cpu-baseline")
    assert "workloads" not in d and d["rotate"] == 4 and d["host_ms_per_step"] > 0
    assert REQUIRED <= set(d) and d["config"]["per_gpu_batch"] == scenes
    assert d["config"]["hip_graphs"] is True and d["value"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_line_is_self_verifying():
    """The N > 1 line carries what is needed to check it from outside: the
    backend and width of the process group, the device every rank drove, and the device time of
    the gradient all-reduce.  Two ranks on the one GPU of this box, gloo between them."""
    port = 36500 + (os.getpid() % 2000)
    env = dict(os.environ, BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", "--no-kernels", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    d = json.loads(lines[0])
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and d["config"]["global_batch"] == 16
    rc = d["rccl"]
    assert rc["backend"] == "gloo" and rc["world_size"] == 2 and rc["ranks_device_ids"] == [0, 0]
    assert rc["ranks_share_one_gpu"] is True and rc["bytes"] > 4_000_000 and rc["in_graph"] is False
    # every step's all-reduce was timed: the feeding loop (1 warm-up + 3 timed + 3 host-time steps
    # + the one that consumes the last prefetch) and the resident loop (1 + 3)
    assert rc["samples"] == 8 + 4 and rc["allreduce_us"] > 0
    assert d["rotate"] == 4 and d["value"] > 0 and d["value_resident"] > 0 and d["host_ms_per_step"] > 0
    # a process group narrower than --gpus is refused
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port + 1),
                          os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0",
                          "--backend", "gloo", "--no-kernels", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert bad.returncode != 0 and "process group of 2 ranks" in bad.stderr


def test_bench_fails_loudly_without_a_gpu():
    """No CPU fallback in the measured path: on a machine without a GPU bench.py stops with a
    clear message instead of timing something else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for GPU-less machines")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and "{" not in r.stdout
