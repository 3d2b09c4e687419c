"""The committed bench lines carry what the bench contract asks for (keys, types, consistency), so a change to bench.py that drops
a field shows up in the CPU suite, not in the driver's parser."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": (int, float),
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


@pytest.mark.parametrize("name", ["r02_bench_n1.json", "r02_bench_n1_bf16.json", "r02_bench_n1_fern.json", "r02_bench_n2_gloo_rehearsal.json",
                                  "r03_bench_n1.json", "r03_bench_n1_bf16.json", "r03_bench_n1_fern.json", "r03_bench_n4_gloo_rehearsal.json",
                                  "r03_bench_rank3_of_8_alone.json", "r03_bench_n1_with_f16_split.json", "r04_bench_n1.json", "r04_bench_n4_gloo_rehearsal.json",
                                  "r05_bench_n1.json", "r05_driver_command_bench_line.json", "r05_bench_n4_gloo_rehearsal.json",
                                  "r05_bench_one_rank_rccl_collective.json", "r05_bench_n1_fern.json", "r05_bench_n1_bf16.json",
                                  "r06_driver_command_bench_line.json", "r06_bench_n1.json", "r06_bench_n1_bf16.json", "r06_bench_n1_fern.json", "r06_bench_n4_both_routes_lego.json", "r06_bench_n4_both_routes_fern.json",
                                  "r06_bench_one_rank_rccl_both_routes.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    path = os.path.join(ROOT, "profiles", name)
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    line = json.loads(lines[-1])
    for key, typ in REQUIRED.items():
        assert key in line and isinstance(line[key], typ), key
    assert "vs_baseline" in line and line["vs_baseline"] is None                  # BASELINE.md publishes no number for this metric
    assert line["unit"] == "rays/s" and line["higher_is_better"] is True and line["scaling"] in ("strong", "weak")
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and 0.0 < roof["frac"] <= 1.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 2e-3
    # whole-job rate: rays of the (strong-scaled) batch per step time
    assert abs(line["value"] - 4096 / (line["ms_per_step"] * 1e-3)) / line["value"] < 2e-3
    if name == "r03_bench_n1.json":                       # the default line carries the bf16 small-batch leg and both weight-gradient legs
        assert [l["rays"] for l in line["bf16"]["small_batch"]] == [256, 512, 1024, 2048]
        assert {"wgrad_256x256", "wgrad_9x256x256"} <= set(line["train"])
        assert line["roofline"]["traffic_is_current"] is True
    if name == "r03_bench_n1_with_f16_split.json":        # the split-precision step is an EXTRA leg: the headline stays the fp32-MFMA kernel's
        leg = line["f16_split"]
        assert line["dtype"] == "f32" and "mlp_fp32_kernel" in roof["kernel"] and roof["peak"] == 157.3
        assert "f16" in leg["dtype"] and leg["rays_per_s"] > 2 * line["value"]
        assert abs(leg["issued_f16_mfma_TFLOPs"] - 3 * leg["network_TFLOPs"]) < 0.5
        assert leg["max_abs_rgb_c_diff_vs_f32"] < 2e-6 and leg["rays_beyond_1e-4_rgb_f"] < 0.005 * 4096
    if name.startswith("r04") and line["n_gpus"] > 1:     # an N > 1 line verifies itself: what the collective saw, from the driver's record alone
        c = line["collective"]
        assert c["world_size"] == line["n_gpus"] and c["backend"] in ("nccl", "gloo")
        assert [r["rank"] for r in c["ranks"]] == list(range(line["n_gpus"])) and all({"host", "device", "name", "pci_bus_id", "cus"} <= set(r) for r in c["ranks"])
        assert 1 <= c["distinct_devices"] <= line["n_gpus"]
        assert c["all_gather_ms"] > 0 and c["all_gather_bytes_assembled"] >= line["frame_hw"][0] * line["frame_hw"][1] * 16
        assert c["frame_equal_across_ranks"] is True and c["neighbour_tile_recomputed_equal"] is True
    if name.startswith("r04") and "solo_rank" not in line:
        assert isinstance(line["frame_checksum"], int)    # the assembled frame's bit patterns: comparable across N
        if line["n_gpus"] > 1:
            assert line["frame_checksum"] == line["collective"]["frame_checksum_rank0"]
    if name.startswith("r04") and line["n_gpus"] == 1:
        assert "collective" not in line                   # the N = 1 line is unchanged
    if name.startswith("r06"):
        with open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8") as f:
            assert line["metric"] == json.load(f)["metric"]                     # byte for byte (U+00D7), read from the file
        if "collective" in line:
            # round 6: ONE run times the frame through both gather routes and the checksums agree; lego's is the N = 1 line's number
            c = line["collective"]
            assert c["tile_gather_route"] == "torch" and "error" not in c["c_abi"]
            assert c["c_abi"]["equal_to_torch_route_on_every_rank"] is True and c["c_abi"]["frame_checksum_equals_torch_route_on_every_rank"] is True
            assert line["frame_ms"] > 0 and line["frame_ms_c_abi"] > 0 and line["frame_checksum_c_abi"] == line["frame_checksum"] == c["frame_checksum_rank0"]
            assert c["world_size"] == line["n_gpus"] and c["c_abi"]["world_size"] == line["n_gpus"]
            if "fern" not in name:
                assert line["frame_checksum"] == 3074984520147328127
            else:
                assert c["c_abi"]["staging_bytes"] > 0                           # 378 rows over 4 ranks: the ragged path
        elif name in ("r06_bench_n1_bf16.json", "r06_bench_n1_fern.json"):
            assert "frame_ms_c_abi" not in line and (line["dtype"] == "bf16") == ("bf16" in name) and ("fern" in line["config"]["workload"]) == ("fern" in name)
        else:
            assert "frame_ms_c_abi" not in line and line["frame_checksum"] == 3074984520147328127
            st = line["staging"]
            assert st["batch_gather_rows"] == 4096 and 0 < st["batch_gather_us"] < 200 and st["materialized_shuffle"]["GBps_algorithmic"] > 1500
            b = line["bf16"]
            assert b["peaked"]["coarse_f16s_fine_bf16"]["psnr_rgb_f_vs_fp32_dB"] > b["peaked"]["bf16"]["psnr_rgb_f_vs_fp32_dB"]
            assert b["coarse_f16s_fine_bf16"]["psnr_rgb_c_vs_fp32_dB"] > 100 and "sample positions" in b["what"].lower()
            assert line["roofline"]["traffic_is_current"] is True
    if name in ("r05_bench_n4_gloo_rehearsal.json", "r05_bench_one_rank_rccl_collective.json"):
        c = line["collective"]                            # round 5: which route assembled the timed frames, and the C ABI's route beside it where RCCL is the backend
        assert c["tile_gather_route"] == "torch" and c["frame_equal_across_ranks"] is True and c["neighbour_tile_recomputed_equal"] is True
        if c["backend"] == "nccl":
            assert c["c_abi"]["equal_to_torch_route_on_every_rank"] is True and c["c_abi"]["all_gather_ms"] > 0
        else:
            assert c["c_abi"] is None and line["frame_checksum"] == 3074984520147328127        # four ranks == one rank, bit for bit
    if name in ("r05_bench_n1.json", "r05_driver_command_bench_line.json"):          # SURVEY 8(d)'s protocol beside the contract's mean: per-step hipEvent median; CPU baseline median of >= 3 reps
        assert abs(line["ms_per_step_median"] - line["ms_per_step"]) < 0.01 * line["ms_per_step"] and "hipEvents" in line["ms_per_step_median_is"]
        assert "median of 3 rep" in line["cpu_baseline"]["sample"]
        assert line["roofline"]["traffic_is_current"] is (name == "r05_bench_n1.json")   # the profiled run preceded the PMC passes that re-tied traffic.json to the build
        assert line["frame_checksum"] == 3074984520147328127          # the frame of rounds 4 and 5, bit for bit (one GPU; an N-GPU line must carry the same)
    if line["n_gpus"] == 1 and "cpu_baseline" in line:
        cpu = line["cpu_baseline"]
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in cpu, key
        assert cpu["kind"] in ("reference", "port")


def test_metric_string_is_baseline_jsons_byte_for_byte():
    """`metric` is read from BASELINE.json (800 U+00D7 800), not retyped: a strict comparer of the two strings accepts the line."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8") as f:
        want = json.load(f)["metric"]
    assert bench.baseline_metric() == want and bench.METRIC_FALLBACK == want and "\u00d7" in want
    assert json.loads(json.dumps({"metric": bench.baseline_metric()}, ensure_ascii=False))["metric"].encode() == want.encode()


def test_worker_exit_status():
    """0 for a clean run; 75 when the C-ABI gather leg hung on the rank; 76 when backend nccl did not see N distinct GPUs (gloo rehearsals on one
    card are exempt: they claim nothing about N GPUs)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.exit_status(False, "nccl", 8, {"distinct_devices": 8}) == 0
    assert bench.exit_status(False, "nccl", 8, {"distinct_devices": 1}) == bench.EXIT_RCCL_SAW_FEWER_GPUS == 76
    assert bench.exit_status(False, "gloo", 4, {"distinct_devices": 1}) == 0
    assert bench.exit_status(False, "nccl", 1, None) == 0
    assert bench.exit_status(True, "nccl", 8, {"distinct_devices": 8}) == bench.EXIT_C_ABI_HUNG == 75
