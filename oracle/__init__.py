"""CPU oracle for the recengine hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A CPU restatement (torch-CPU fp32 for the differentiable math, C with `fmaf`
for the bit-exact scoring/top-K order, numpy for index work) of the arithmetic
the reference executes on the hot path named in BASELINE.json `north_star`
(SURVEY.md §8a):

  oracle.embedding  gather_rows / scatter_add_rows / sasrec_embed    SASRec/main.py:178-187, MF-BPR/main.py:84-86
  oracle.criterions BPR / BCE / CE / regularize                      call sites MF-BPR/main.py:88-91, SASRec/main.py:205-219,
                                                                     LightGCN/main.py:95-106, DeepFM/main.py:214
  oracle.sasrec     encode / fit / recommend_from_full               SASRec/main.py:31-50,159-228
  oracle.mf         MF-BPR fit / recommend_from_full                 MF-BPR/main.py:78-109
  oracle.lightgcn   encode / fit / full scores                       LightGCN/main.py:77-125
  oracle.deepfm     encode / fit / recommend_from_pool               DeepFM/main.py:58-62,80-85,119-124,201-219
  oracle.ranking    score + seen-mask + top-K + metrics              Coach.evaluate mirror UniSRec/main.py:400-447
  oracle.adam       dense Adam with coupled L2                       cfg dump benchmark/Amazon2014Beauty_550_LOU/SASRec.json:254-300
  oracle.c/         C restatement (fmaf chain scoring, top-K, gather, scatter) for larger sizes

Who may import this package: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` -- as the checker / the timed CPU baseline,
never as a product path.  `recboard_amd/` must not import it.

Pinning status
--------------
* The torch-visible math (everything in the reference's `main.py` files) is
  PINNED: `tests/test_oracle_golden.py` checks every function here against
  golden vectors produced by importing the reference classes themselves
  (`tests/golden/make_golden.py`, run in the development container).
* The arithmetic that lives in the third-party package `freerec` (pinned 1.0.1
  by `freerec.declare`, e.g. SASRec/main.py:7; NOT vendored, NOT installable
  offline) is **parity unpinned**: BPRLoss/BCELoss4Logits/CrossEntropy4Logits
  reductions, `regularize`, the metric definitions (HITRATE/NDCG/...), the
  seen-mask constant (-1e23, visible only in the copy at UniSRec/main.py:413),
  `to_normalized_adj`.  They are restated from the reference's call sites and
  from internal-consistency checks; the goldens for them were
  produced with the same restatement (tests/golden/_freerec_standin.py).
"""
