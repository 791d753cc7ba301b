"""`freerec.parser.Parser`: module-level `cfg = Parser(); cfg.add_argument(...); cfg.set_defaults(...); cfg.compile()`
(SASRec/main.py:9-28).  Precedence as the scripts rely on: built-in defaults < set_defaults < the yaml given by --config < command line.
Every option is an attribute (`cfg.embedding_dim`); `cfg.get(key, default)`; both spellings of the optimizer options
(`optim_*_moment_decay` as LightGCN/main.py:136-150 reads them, `adam_beta*` / `sgd_*` as DeepFM/main.py:229-246 does)."""
import argparse
import sys

CORE_DEFAULTS = dict(
    root="../../data", dataset=None, tasktag=None, config=None, ranking="full", retain_seen=False,
    device="cuda:0", ddp_backend="nccl", num_workers=4, pin_memory=False,
    optimizer="adam", lr=1e-3, weight_decay=0.0, nesterov=False, momentum=0.9, beta1=0.9, beta2=0.999,
    optim_first_moment_decay=0.9, optim_second_moment_decay=0.999, adam_beta1=0.9, adam_beta2=0.999, sgd_momentum=0.9, sgd_nesterov=False,
    lr_scheduler={}, batch_size=256, epochs=None, seed=1, eval_freq=5, eval_valid=True, eval_test=False, early_stop_patience=1e23,
    monitors=["LOSS", "HitRate@1", "HitRate@5", "HitRate@10", "NDCG@5", "NDCG@10"], which4best="NDCG@10",
    description="RecSys", id=None, resume=False, log2file=True, log2console=True, checkpoint_path=None, log_path=None,
    engine="auto",   # recengine: "auto" routes recognised models onto the fused kernels, "module" keeps the script's own torch code
)


class Parser:
    def __init__(self):
        object.__setattr__(self, "_ap", argparse.ArgumentParser(allow_abbrev=False))
        object.__setattr__(self, "_opts", dict(CORE_DEFAULTS))
        object.__setattr__(self, "_compiled", False)
        for k, v in CORE_DEFAULTS.items():
            flag = "--" + k.replace("_", "-")
            if isinstance(v, bool):
                self._ap.add_argument(flag, type=lambda s: str(s).lower() in ("1", "true", "yes"), default=None)
            elif isinstance(v, (list, dict)):
                self._ap.add_argument(flag, type=str, default=None)
            else:
                self._ap.add_argument(flag, type=(type(v) if v is not None and not isinstance(v, float) else (float if isinstance(v, float) else str)), default=None)

    # -- the script's own options: remembered with their defaults, parsed in compile()
    def add_argument(self, *flags, **kw):
        dest = kw.get("dest") or flags[-1].lstrip("-").replace("-", "_")
        self._opts[dest] = kw.get("default")
        kw = dict(kw)
        kw["default"] = None
        try:
            self._ap.add_argument(*flags, **kw)
        except argparse.ArgumentError:          # a core option re-declared by the script: its default wins
            pass

    def set_defaults(self, **kw):
        self._opts.update(kw)

    def compile(self):
        ns, _ = self._ap.parse_known_args(sys.argv[1:])
        cli = {k: v for k, v in vars(ns).items() if v is not None}
        path = cli.get("config") or self._opts.get("config")
        if path:
            import yaml
            with open(path) as f:
                self._opts.update({str(k).replace("-", "_"): v for k, v in (yaml.safe_load(f) or {}).items()})
        for k, v in cli.items():
            if k == "monitors" and isinstance(v, str):
                v = [s for s in v.replace(",", " ").split() if s]
            self._opts[k] = v
        o = self._opts
        # the two spellings of the moment / momentum options follow each other unless both were given
        for a, b in (("optim_first_moment_decay", "adam_beta1"), ("optim_second_moment_decay", "adam_beta2")):
            if a in cli and b not in cli:
                o[b] = o[a]
            elif b in cli and a not in cli:
                o[a] = o[b]
        object.__setattr__(self, "_compiled", True)
        return self

    def get(self, key, default=None):
        return self._opts.get(key, default)

    def __getattr__(self, key):
        try:
            return self.__dict__["_opts"][key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key, value):
        self._opts[key] = value

    def __contains__(self, key):
        return key in self._opts

    def to_dict(self):
        return dict(self._opts)
