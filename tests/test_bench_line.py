"""The driver parses the LAST stdout line of bench.py: it must stay compact whatever rides in the full record (round 5's line had
grown to 25 KB and could not be parsed).  CPU test: builds the line from a canned full record."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _canned():
    long_note = "x" * 3000
    per_rank = [dict(rank=r, transport="peer", peer_self_check=dict(detail=long_note), status_word=0) for r in range(8)]
    return dict(
        metric="particle_steps_per_sec", value=1.85e10, unit="particle-steps/s", n_gpus=8, steps=200, warmup=20, ms_per_step=0.0566,
        higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
        config=dict(workload="gmm_c8_d16 ImportanceK: propagate+reweight+LSE, systematic resample, gather (BASELINE.json configs[1])",
                    k_particles_per_gpu=1 << 20, k_particles_total=1 << 23, rng_stream="flat", sharding="particles x8", exchange="peer",
                    exchange_stats=dict(transport="peer", ranks=8, status=0, note=long_note, per_rank=per_rank, transports_agree=True,
                                        any_status_bit=False)),
        roofline=dict(bound="hbm", kernel="gjx::k_run_gmm_flat<16,4,256>", achieved=3480.0, peak=8000.0, unit="GB/s", frac=0.435, traffic=None,
                      traffic_from_profiles=dict(bytes_per_launch=80.2e6, file="profiles/r05_pmc_traffic.json", note=long_note),
                      stream=long_note, kernel_us=22.9, timing=long_note, algorithmic_bytes_per_launch=79691776, launches_per_step=2,
                      note=long_note, back_to_back=dict(kernel_us=23.8, frac=0.419, note=long_note)),
        log_ml=-43.8, log_ml_exact=-43.81, log_ml_rel_err=2e-5, timing_note=long_note,
        roofline_jax32_stream=dict(kernel="k", kernel_us=73.6, frac=0.135),
        cpu_baseline=dict(value=1.05e7, unit="particle-steps/s", cores=16, kind="port", build=long_note, sample=long_note,
                          single_thread=dict(value=1e6, sample=long_note)),
        jit=dict(hiprtc_compiles=0, disk_hits=3, hiprtc_ms=0.0, structures=3),
        extra=dict(ssm=dict(value=2.3e10, unit="particle-steps/s", ms_per_step=2.9, log_ml_rel_err=4e-5, log_ml_z=0.4,
                            config=dict(workload="lgssm " + long_note), roofline=dict(bound="hbm", kernel="k_pf_persistent " + long_note, frac=0.26, note=long_note)),
                   hmc=dict(value=1.7e9, unit="chain-leapfrogs/s", ms_per_step=38.7, accept_rate=0.93, config=dict(workload="hmc"),
                            roofline=dict(bound="mfma", kernel="k_hmc_logreg_mfma2", frac=0.82, note=long_note)),
                   round5={"blob%d" % i: long_note for i in range(10)}),
    )


def test_headline_line_is_compact_and_complete():
    import bench
    res = _canned()
    assert len(json.dumps(res)) > 60_000                    # the full record is as unwieldy as round 5's
    line = bench.headline_line(res, os.path.join(ROOT, "bench_extra.json"))
    assert len(line) < bench.HEADLINE_MAX_BYTES < 8192 and "\n" not in line
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "log_ml_rel_err", "jit_compiles_at_runtime"):
        assert k in rec, k
    assert rec["roofline"]["frac"] == 0.435 and rec["roofline"]["bound"] == "hbm" and rec["roofline"]["peak"] == 8000.0
    assert set(rec["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert rec["cpu_baseline"]["value"] == 1.05e7 and rec["cpu_baseline"]["cores"] == 16 and rec["cpu_baseline"]["kind"] == "port"
    assert "extra" not in rec and "exchange_stats" not in rec["config"] and rec["config"]["exchange_summary"]["transports_agree"] is True
    assert rec["other_configs"]["ssm"]["roofline"]["frac"] == 0.26 and rec["other_configs"]["ssm"]["log_ml_z"] == 0.4
    assert rec["jit_compiles_at_runtime"] == 0 and rec["full_record"] == "bench_extra.json"


def test_headline_line_survives_oversized_optional_blocks():
    import bench
    res = _canned()
    res["config"]["sharding"] = "y" * 190
    res["config"].update({"k%d" % i: "z" * 190 for i in range(30)})       # a config that alone would pass the limit
    line = bench.headline_line(res, None)
    rec = json.loads(line)
    assert len(line) < bench.HEADLINE_MAX_BYTES and rec["roofline"]["frac"] == 0.435 and rec["cpu_baseline"]["value"] == 1.05e7
