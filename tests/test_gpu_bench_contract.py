"""bench.py's output contract (one JSON line; metric / value / roofline / cpu_baseline fields) on a short clip."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly ONE line on stdout
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run("--frames", "12", "--steps", "2", "--warmup", "1", "--end-to-end")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["unit"] == "frames/s" and d["value"] > 0
    assert abs(d["value"] - d["config"]["encoded_frames_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    assert "not cv::dct" in c["dct_leg"] and c["dct_is_reference"] is False
    # two rows beside each other: the configuration's own level count and the reference's default SSE2 4-level build,
    # each on one core and on all cores, each leg of the CPU frame timed separately
    rows = c["rows"]
    assert set(rows) == {"config", "sse2_4level"} and rows["config"]["levels"] == 3 and rows["sse2_4level"]["levels"] == 4
    for row in rows.values():
        assert row["value"] > 0 and row["cores"] == 1 and row["hbma_ms_per_frame"] > 0 and row["dct_ms_per_frame"] > 0
        assert row["all_cores"] is None or (row["all_cores"]["value"] > 0 and row["all_cores"]["cores"] >= 1)
    assert c["value"] == rows["config"]["value"]
    # PCIe-inclusive rates of the host-facing ways in, outside the timed region: SCHEMA only.  No
    # inequality on a wall-clock figure lives in this file (a 12-frame, 2-step run on a cold box measures start-up; such a ratio
    # voided round 4's whole GPU suite): performance expectations are tools/perf_expectations.py, which gates nothing.
    e = d["end_to_end"]
    assert e["pcie_inclusive"] is True and e["unit"] == "frames/s" and e["bound"]
    for key in ("reference_signatures_fps", "stream_encoder_fps", "reference_application_fps", "reference_application_batched_encoder_fps"):
        assert key in e and (e[key] is None or e[key] > 0), (key, e)  # None = not measured here (the *_note beside it says why)
        assert e[key] is not None or key.replace("_fps", "_note") in e, (key, e)
    # the batched driver's rate explains itself (round 6): >= 1 s of passes, per-batch phases from the encoder's own clocks, the cores it had
    if e["stream_encoder_fps"] is not None:
        ph = e["stream_encoder_phases"]
        assert ph["seconds"] >= 1.0 or ph["passes"] == 64
        assert set(ph["host_ms_per_batch"]) == {"staging", "slot_wait", "deliver_wait", "sink", "wall"} and set(ph["device_ms_per_batch"]) == {"h2d", "kernels", "d2h"}
        assert ph["host_cores"] >= 1 and ph["copy_threads"] >= 1 and ph["batches"] >= ph["passes"] >= 1 and ph["h2d_GBps"] > 0 and ph["d2h_GBps"] > 0
        assert len(ph["d2h_GBps_by_pass"]) == ph["passes"] and min(ph["d2h_GBps_by_pass"]) > 0  # (a step in it = another process's idle copy queues)
        # ... and the same frames as ONE stream (a single Encode call): as many encoded frames as the passes had, the same phase keys
        ls = e["stream_encoder_long_stream"]
        assert ls["encoded_frames"] >= 64 * ph["passes"] and ls["frames_per_s"] > 0 and ls["batches"] >= ph["passes"]
        assert set(ls["host_ms_per_batch"]) == set(ph["host_ms_per_batch"]) and set(ls["device_ms_per_batch"]) == set(ph["device_ms_per_batch"])
    # what `value` is, and what a clip encoded once costs beside it (round 6): schema and identities only
    assert d["config"]["value_is"].startswith("steady state") and d["config"]["chunks_per_step"] >= 1 and d["config"]["output_sets"] >= 1
    pol = d["config"]["speculation_policy"]
    assert pol["chunk_launches"] == d["steps"] * d["config"]["chunks_per_step"] and 0 <= pol["speculated"] <= pol["had_the_choice"] <= pol["chunk_launches"]
    fe = d["first_encode"]
    assert fe["unit"] == "frames/s" and fe["reps"] >= 1 and fe["encoded_frames"] == d["config"]["encoded_frames_per_step"]
    for key in ("once_through", "with_prior"):
        row = fe[key]
        assert row["ms_min"] <= row["ms_median"] <= row["ms_max"] and abs(row["value"] - fe["encoded_frames"] / (row["ms_median"] * 1e-3)) <= 1e-6 * row["value"]
        # (a step into an empty pipeline may run in two chunks whatever chunks_per_step says: the idle-pipeline rule on big shards)
        assert 0 <= row["chunk_launches_speculated"] <= row["chunk_launches"] and row["chunk_launches"] in (fe["reps"] * fe["chunks_per_step"], 2 * fe["reps"])
    # a load voids the policy: a once-through step speculates only blind, on the second half of the mixed form (big shards)
    ot = fe["once_through"]
    assert ot["chunk_launches_speculated"] == 0 or d["config"]["chunks_per_step"] > 1 or (ot["chunk_launches"] == 2 * fe["reps"] and ot["chunk_launches_speculated"] == fe["reps"])
    assert fe["policy_voided_steps"]["steps"] == d["steps"] and fe["policy_voided_steps"]["value"] > 0
    # a stream of DIFFERENT clips, each encoded once per rotation, stepped where they are (svc_clip_step_frames): schema and identities
    sc = fe["stream_of_clips"]
    assert sc is not None, fe.get("stream_of_clips_note")
    assert sc["distinct_clips"] == min(d["steps"], 24) and sc["steps"] == d["steps"] and abs(sc["value"] - fe["encoded_frames"] / (sc["ms_per_clip"] * 1e-3)) <= 1e-6 * sc["value"]
    assert 0 <= sc["chunk_launches_speculated"] <= sc["chunk_launches"]
    assert set(d["hbm_streaming_measured"]) >= {"read_only", "write_only", "copy_1_read_1_write", "3_read_1_write", "unit"}
    assert not any(k.startswith("frac_of_streaming") for k in r)
    assert "traffic_source" in r and d["config"]["driver"].startswith("svc::ClipEncoder")
    # counter traffic is tied to the build (schema, both branches): a number with the file it came from, or null with the reason
    for rf in (r, d["roofline_dct"]):
        assert (rf["traffic"] is None and rf["traffic_source"].startswith("null: ")) or \
               (rf["traffic"] > 0 and rf["traffic_source"].startswith("offline PMC (profiles/")), rf
    # the main stream, back to back; type_patch = the foreground tiles redone once the region ids exist (one pass over the BGR clip)
    assert set(d["kernel_ms_per_step"]) == {"luma_pyramid", "hbma", "dct_quant"}
    assert 0.0 <= d["config"]["foreground_mv_blocks"] <= 1.0
    # beside it (pipelined schedule); type_patch: what a step that read the BGR clip once owes after its segmentation (none before the
    # speculation policy has a foreground share)
    assert set(d["overlapped_ms_per_step"]) - {"type_patch"} == {"ransac", "segment", "note"}
    assert all(v > 0 for v in d["kernel_ms_per_step"].values())
    # the sustained loop behind the timed region (back-to-back steps, untimed for `value`): identities only
    s = d["sustained"]
    assert s["steps"] > 0 and s["steps"] % 50 == 0 and s["seconds"] > 0 and d["sustained_ms_per_step"] == s["ms_per_step"]
    assert abs(s["ms_per_step"] - s["seconds"] / s["steps"] * 1e3) <= 0.05 * s["ms_per_step"]
    assert abs(s["value"] - d["config"]["encoded_frames_per_step"] / (s["ms_per_step"] * 1e-3)) <= 1e-6 * s["value"]


def test_single_rank_under_the_launcher_is_the_plain_run():
    """The scaling driver launches N = 1 through torch.distributed.run like every other N: same code path, same line
    (world size 1: no process group, no halo, `scaling` weak) -- so SCALE's N = 1 agrees with BENCH by construction."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", "12", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-hbm-probe", "--sustain-seconds", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    launched = json.loads(lines[0])
    plain = _run("--gpus", "1", "--frames", "12", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-probe", "--sustain-seconds", "0")
    assert launched["n_gpus"] == plain["n_gpus"] == 1 and launched["scaling"] == plain["scaling"] == "weak"
    assert launched["config"] == plain["config"] and "multi_gpu" not in launched and "sustained" not in launched
    assert set(launched) == set(plain) and set(launched["kernel_ms_per_step"]) == set(plain["kernel_ms_per_step"])


def test_bench_other_config_and_flags():
    d = _run("--config", "C2-720p-3L-dct8", "--frames", "6", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--wire")
    assert "cpu_baseline" not in d and d["config"]["workload"] == "C2-720p-3L-dct8" and d["value"] > 0


def test_bench_serial_schedule():
    d = _run("--frames", "10", "--steps", "3", "--warmup", "4", "--no-cpu-baseline", "--schedule", "serial", "--no-hbm-probe")
    assert "roofline" in d and "one stream" in d["config"]["schedule"]
    assert set(d["kernel_ms_per_step"]) - {"type_patch"} == {"luma_pyramid", "hbma", "ransac", "segment", "dct_quant"} and "overlapped_ms_per_step" not in d


def test_bench_two_ranks_on_one_gpu_rehearsal():
    """N = 2 end to end on ONE device (gloo rehearsal switch; RCCL refuses two ranks on a GPU): strong sharding of
    the clip, halo through the encoder's transport hook, the halo self-check, the weak figure as second field."""
    env = dict(os.environ, SVC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29513", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "9", "--steps", "3",
                        "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["clip_frames"] == 9
    assert d["config"]["frames_per_gpu"] == [5, 4] and d["config"]["encoded_frames_per_step"] == 8
    assert d["weak"]["clip_frames"] == 18 and d["weak"]["encoded_frames_per_step"] == 17 and d["weak"]["value"] > 0
    assert d["halo_exchange_ms"] is not None
    # the self-diagnosis of a multi-rank run: which transport carried the halo, what the communicator says about
    # itself (none here: gloo rehearsal), the halo checksum verdict, every rank's own clock and halo time
    m = d["multi_gpu"]
    assert "gloo" in m["transport"] and m["rccl_ranks"] is None
    assert m["halo_check"]["verdict"] == "ok" and m["halo_check"]["ranks_checked"] == 1
    assert len(m["ms_per_step_by_rank"]) == 2 and m["ms_per_step_min"] == min(m["ms_per_step_by_rank"]) and m["ms_per_step_max"] == max(m["ms_per_step_by_rank"])
    assert m["frames_by_rank"] == [5, 4] and m["encoded_by_rank"] == [4, 4]
    assert len(m["halo_exchange_ms_by_rank"]) == 2 and all(v is not None and v > 0 for v in m["halo_exchange_ms_by_rank"])
    # which order each rank ran (round 6): the speculation policy is per rank and involves no collective, so the line carries, per rank, the
    # launches of the timed region that had the choice and those that speculated (a 5-frame shard never has the choice: two passes)
    pr = m["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and [x["frames"] for x in pr] == [5, 4] and [x["encoded"] for x in pr] == [4, 4]
    assert all(x["ms_per_step"] > 0 and x["launches_speculated"] <= x["launches_with_the_choice"] for x in pr)
    assert all(x["bgr_passes_per_step"] in ("two", "one", "mixed") for x in pr) and [x["bgr_passes_per_step"] for x in pr] == ["two", "two"]
    # the prediction DESIGN.md section 6 makes for this shard size sits next to the measurement (none for a 5-frame shard:
    # the committed table has the BASELINE shard sizes; the field says so instead of inventing a number)
    p = m["prediction"]
    assert "predicted_ms_per_step" in p and (p["predicted_ms_per_step"] is None) == ("note" in p)


def test_bench_halo_check_failure_is_collective():
    """A halo that does not arrive intact must end EVERY rank non-zero (a one-sided exit would leave the others in the next
    barrier until the launcher times out), naming the failing rank."""
    env = dict(os.environ, SVC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    # bench.py itself carries no fault hook: tests/helpers/bench_corrupt_halo.py wraps its halo transport and runs its main()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", os.path.join(ROOT, "tests", "helpers", "bench_corrupt_halo.py"), "--gpus", "2", "--frames", "9", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "halo self-check failed on rank(s) [1]" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must not run one rank and print n_gpus 1 (round 4 did, silently): the
    process becomes the launcher (torch.distributed.run as a CHILD, the parent never touches the GPU) and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SVC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")  # rehearsal switch: two ranks share this box's one GPU
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "9", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--scaling", "strong"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["frames_per_gpu"] == [5, 4] and len(d["multi_gpu"]["ms_per_step_by_rank"]) == 2
    assert d["multi_gpu"]["launcher"].startswith("bench.py itself") and d["multi_gpu"]["rccl_ranks"] is None  # gloo rehearsal
    assert "starting 2 ranks as a child" in r.stderr
