"""CPU oracle for the HSTU / multi-head decode hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`multi-head-recommendation-with-human-priors_amd/`) imports this package; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may.

The oracle is a from-scratch fp32 restatement (torch CPU tensors + numpy) of the
reference's algorithm for the path named in BASELINE.json `north_star`:

  * `hstu_oracle`   - HSTU encoder, decoding heads, sampled-softmax (nce) and
                      prior losses, `predict`, `compute_item_all`
                      (reference `code/REC/model/IDNet/hstu.py`)
  * `decode_oracle` - per-head top-k, cross-head merge + dedup, hit matrix
                      (reference `code/REC/evaluator/collector.py`), history
                      suppression (`code/REC/trainer/trainer.py:724-726`)
  * `metrics_oracle`- Recall / NDCG / Entropy sums
                      (reference `code/REC/evaluator/metrics.py`)
  * `optim_oracle`  - dense AdamW step + cosine warm-up schedule
                      (reference `code/REC/trainer/trainer.py:292-299`,
                      `code/REC/utils/lr_scheduler.py:79-116`)

Parity pin: the reference has no tests or golden vectors of its own (SURVEY.md
section 4).  The oracle is pinned against outputs of the reference itself,
produced in the build container by `tests/gen_golden.py` (which imports the
reference's own `hstu.py` / `collector.py` / `metrics.py` from /root/reference)
and committed as `.npz` fixtures under `tests/golden/`.
`tests/test_oracle_golden.py` checks every oracle function against them.
"""
