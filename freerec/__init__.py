"""`freerec`-compatible surface of the recengine (the builder's own code -- NOT the third-party FreeRec package).

RecBoard's model scripts are thin clients of `freerec` 1.0.1 (`freerec.declare(version="1.0.1")`, SASRec/main.py:7), which is not
vendored in the reference and does not ship to the GPU box.  This package answers to the names those scripts touch (SURVEY.md
Appendix B: `declare`, `parser.Parser`, `data.{tags,fields,datasets}`, `models.{Gen,Seq,Pred}RecArch`, `criterions`, `launcher.Coach`,
`graph`, `ddp`, `utils`), written from the scripts' call sites, so that a `main.py` of the reference imports and runs unchanged:

    PYTHONPATH=/root/repo python /root/reference/SASRec/main.py --config=configs/x.yaml

and routes the hot path onto the engine: `launcher.Coach` recognises a SASRec-shaped module and drives it through the fused training
step (`recboard_amd.sasrec.SASRecEngine`: one batch-preparation launch + one hipGraph replay per step) with the module's parameters
living in the engine's arena; evaluation goes through the fused score + mask + top-K kernel.  Everything else runs as ordinary torch
modules (on the `recengine::*` custom ops where the module uses them).

Parity note: FreeRec's own arithmetic -- criterion reductions, `regularize`, metric definitions, sampler
distributions, `to_normalized_adj` -- is restated from call sites, not from its source: "parity unpinned" at that boundary.
"""
__version__ = "1.0.1"


def declare(version: str = __version__):
    """`freerec.declare(version=...)`: the scripts pin the FreeRec version they were written for (SASRec/main.py:7)."""
    want, have = str(version).split(".")[:2], __version__.split(".")[:2]
    if want != have:
        import warnings
        warnings.warn(f"this script declares freerec {version}; the recengine surface follows the {__version__} call sites")


from . import criterions, data, ddp, graph, launcher, metrics, models, parser, utils  # noqa: E402,F401
