"""Row (e) / a22 on the real model with world size 2: two processes on ONE GPU (gloo host group, gradient buckets through
torch.distributed), started by tests/conftest.py before the session touched the device; this file reads their reports
(tests/ddp_two_ranks_worker.py says what each rank establishes).

Reference protocol: fairseq/distributed/legacy_distributed_data_parallel.py:76-160 (every gradient = sum over ranks / world),
fairseq/trainer.py:714-741 (all-reduce, then x world / sample_size, clip, step)."""
import json
import os
import time

import pytest

pytestmark = pytest.mark.gpu

VARIANTS = ("transformer_fp32", "conformer_bf16")


@pytest.fixture(scope="module")
def reports(ddp_two_ranks):
    if not ddp_two_ranks["procs"]:
        pytest.skip("the two ranks were not started: %s" % ddp_two_ranks["skipped"])
    out = ddp_two_ranks["dir"]
    deadline = time.time() + 900
    for p in ddp_two_ranks["procs"]:
        p.wait(timeout=max(1.0, deadline - time.time()))
    res = []
    for rank in range(2):
        path = os.path.join(out, "rank%d.json" % rank)
        log = open(os.path.join(out, "rank%d.log" % rank)).read()[-4000:]
        assert os.path.exists(path), "rank %d left no report; its log ends:\n%s" % (rank, log)
        r = json.load(open(path))
        assert r.get("ok"), "rank %d failed:\n%s\n%s" % (rank, r.get("error"), log)
        res.append(r)
    return res


@pytest.mark.parametrize("variant", VARIANTS)
def test_reduced_gradient_is_the_mean_of_the_single_process_gradients(reports, variant):
    for r in reports:
        v = r[variant]
        print("rank %d %s: %d buckets, mean_err %s, launched before the end %s, repeat noise %.2e" % (
            r["rank"], variant, v["buckets"], v["mean_err"], v["launched_before_the_end"], v["repeat_noise"]))
        assert v["buckets"] >= 8
        assert v["ranks_differ"] > 0.1  # the two ranks really hold different batches
        # learning pass (all buckets reduced at the end) and the two overlapped passes: the mean, to fp32 summation noise
        # (two identical single-process passes differ by repeat_noise through the fp32 atomics of the per-parameter sums)
        for e in v["mean_err"]:
            assert e <= max(1e-6, 4 * v["repeat_noise"]), v["mean_err"]
        assert v["launched_before_the_end"][0] == 0  # nothing is launched while the ready counts are being learned
        for n in v["launched_before_the_end"][1:]:  # then buckets go out from the hooks, inside backward (its end callback included)
            assert n >= 1, (n, v["buckets"])
        assert v["reduced_equal_on_ranks"]


@pytest.mark.parametrize("variant", VARIANTS)
def test_every_parameter_reports_ready_the_same_number_of_times(reports, variant):
    for r in reports:
        v = r[variant]
        assert v["ready_counts_stable"] and v["ready_counts_equal_on_ranks"]
        # every parameter of the model reports at least once (a parameter that never reports would hold its bucket back until
        # all_reduce_grads; one that reports early would send the bucket out without its last contribution — the mean test)
        assert v["silent_with_gradient"] == [], v["silent_with_gradient"]
        assert v["ready_params"] >= 0.9 * v["params"], (v["ready_params"], v["params"])
        tied = {k: c for k, c in v["ready_counts"].items() if c > 1}
        print("rank %d %s: %d parameters, reporting more than once: %s" % (r["rank"], variant, v["params"], tied))


@pytest.mark.parametrize("variant", VARIANTS)
def test_three_captured_updates_follow_the_eager_data_parallel_trajectory(reports, variant):
    for r in reports:
        v = r[variant]
        print("rank %d %s: moved %.3e, eager vs captured mean diff %.3e, max rel %.3e" % (
            r["rank"], variant, v["moved"], v["traj_mean_diff"], v["traj_max_diff_rel"]))
        assert v["moved"] > 1e-4
        assert v["traj_mean_diff"] <= 0.02 * v["moved"], (v["traj_mean_diff"], v["moved"])
        le, lg = v["losses_eager"], v["losses_graph"]
        for a, b in zip(le[3:], lg[3:]):
            assert abs(a - b) <= 2e-3 * abs(a), (le, lg)
        # the replicas stay together: equal reduced gradients (asserted above, bit for bit), and a clip coefficient from each rank's
        # own fp32 atomic sum of squares — the last bit of it may differ, nothing else
        assert v["masters_rank_diff_eager"] <= 1e-6 and v["masters_rank_diff_graph"] <= 1e-6, (
            v["masters_rank_diff_eager"], v["masters_rank_diff_graph"])
