"""Host part of the engine-adoption decision (recboard_amd/bridge.py: `survey`) on CPU: the model scripts -- the reference's own files
IMPORTED IN PLACE from /root/reference (never copied; skipped where the reference is absent, i.e. on the GPU box) and this repo's
examples/ -- are recognised by structure, the optimizer the script built is read (LightGCN: no decay; DeepFM: the two decay groups), the
script's own `train_per_epoch` is run on one batch with its optimizer / scheduler calls recorded, and the gradients it hands to
`optimizer.step()` are the reference's golden gradients.  What is left to the GPU tests is the engine's side of the comparison."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_freerec_compat import G, REF, ROOT, import_script, needs_ref, toy_dataset  # noqa: E402


def _cfg(mod, **over):
    cfg = mod.cfg
    cfg.device, cfg.engine, cfg.epochs, cfg.eval_freq = "cpu", "auto", 1, 1
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def _mf_case(mod, coach_name):
    z = np.load(os.path.join(G, "mfbpr.npz"))
    U, N = z["param/User.embeddings.weight"].shape[0], z["param/Item.embeddings.weight"].shape[0]
    ds = toy_dataset(U, N)
    model = mod.MF(ds)
    model.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}, strict=True)
    batch = {model.User: torch.from_numpy(z["in/users"]), model.IPos: torch.from_numpy(z["in/pos"]), model.INeg: torch.from_numpy(z["in/neg"]),
             model.Size: len(z["in/users"])}
    cfg = _cfg(mod, monitors=["LOSS"], which4best="LOSS")
    coach = getattr(mod, coach_name)(dataset=ds, trainpipe=[batch], validpipe=None, testpipe=None, model=model, cfg=cfg)
    return z, model, coach


def lightgcn_dataset(zl):
    import freerec
    U, N = zl["param/User.embeddings.weight"].shape[0], zl["param/Item.embeddings.weight"].shape[0]
    crow, col = zl["in/adj_crow"], zl["in/adj_col"]
    rows = np.repeat(np.arange(len(crow) - 1), np.diff(crow))
    eu, ei = rows[rows < U], col[rows < U] - U
    e = np.zeros(0, np.int64)
    return freerec.data.datasets.RecDataSet.from_splits((eu, ei), (e, e), (e, e), U, N)


def _lightgcn_case(mod, coach_name):
    z = np.load(os.path.join(G, "lightgcn.npz"))
    ds = lightgcn_dataset(z)
    model = mod.LightGCN(ds)
    model.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}, strict=False)
    batch = {model.User: torch.from_numpy(z["in/users"]), model.IPos: torch.from_numpy(z["in/pos"]), model.INeg: torch.from_numpy(z["in/neg"]),
             model.Size: len(z["in/users"])}
    cfg = _cfg(mod, monitors=["LOSS"], which4best="LOSS", weight_decay=float(z["cfg/weight_decay"]))
    coach = getattr(mod, coach_name)(dataset=ds, trainpipe=[batch], validpipe=None, testpipe=None, model=model, cfg=cfg)
    return z, model, coach


def deepfm_dataset(z, rows=0):
    import freerec
    from freerec.data import tags as T
    F = freerec.data.fields.Field
    counts = z["cfg/counts"].tolist()
    fields = [F(f"F{i}", T.FEATURE, T.SPARSE, T.EMBED, *((T.USER, T.ID) if i == 0 else (T.ITEM, T.ID) if i == 1 else ()), count=c)
              for i, c in enumerate(counts)]
    fields.append(F("LABEL", T.LABEL))
    cols = {f"F{i}": z["in/x"][:rows, i] for i in range(len(counts))}
    cols["LABEL"] = z["in/labels"][:rows, 0]
    return freerec.data.datasets.PredictionRecDataSet.from_columns(cols, cols, cols, fields)


def load_deepfm_golden(model, z):
    with torch.no_grad():
        for i, f in enumerate(model.input_fields):
            f.embeddings.weight.copy_(torch.from_numpy(z[f"table/{i}"]))
            f.embeddings_lr.weight.copy_(torch.from_numpy(z[f"table_lr/{i}"]))
        model.fm.lr_layer.bias.copy_(torch.from_numpy(z["param/fm.lr_layer.bias"]))
        model.dnn.load_state_dict({k[len("param/dnn."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/dnn.")})


def _deepfm_case(mod, coach_name):
    z = np.load(os.path.join(G, "deepfm.npz"))
    ds = deepfm_dataset(z)
    model = mod.DeepFM(ds)
    load_deepfm_golden(model, z)
    batch = {f: torch.from_numpy(z["in/x"][:, i:i + 1]) for i, f in enumerate(model.input_fields)}
    batch[model.Label] = torch.from_numpy(z["in/labels"])
    batch[model.Size] = len(z["in/labels"])
    cfg = _cfg(mod, monitors=["LOSS", "LOGLOSS", "AUC"], which4best="AUC", eval_freq=1)
    coach = getattr(mod, coach_name)(dataset=ds, trainpipe=[batch], validpipe=None, testpipe=None, model=model, cfg=cfg)
    return z, model, coach


DEEPFM_ARGS = ["--hidden-dims", "32,24,16", "--batch-norm", "True", "--hidden-dropout-rate", "0.3"]
SCRIPTS = [pytest.param(REF, marks=needs_ref, id="reference"), pytest.param(os.path.join(ROOT, "examples"), id="examples")]


@pytest.mark.parametrize("root", SCRIPTS)
def test_mfbpr_script_is_surveyed_as_mf_with_the_golden_step_gradients(root):
    from recboard_amd import bridge
    mod = import_script(os.path.join(root, "MF-BPR", "main.py"), "_bridge_mf_" + str(abs(hash(root))), [])
    z, model, coach = _mf_case(mod, "CoachForMFBPR" if root == REF else "CoachForMF")
    before = {k: v.clone() for k, v in model.state_dict().items()}
    cls, spec, plan, grads, sched = bridge.survey(coach)
    assert cls is bridge.MFAdapter and sched is None
    assert spec.lr == mod.cfg.lr and set(spec.wd.values()) == {mod.cfg.weight_decay} and plan["weight_decay"] == mod.cfg.weight_decay
    for k in ("User.embeddings.weight", "Item.embeddings.weight"):
        np.testing.assert_allclose(grads[k].numpy(), z["grad/" + k], rtol=1e-4, atol=1e-8)
    for k, v in model.state_dict().items():                    # the probe put everything back
        assert torch.equal(v, before[k])
    assert len(coach.optimizer.state) == 0 and coach._meters["train"]["LOSS"].n == 0


@pytest.mark.parametrize("root", SCRIPTS)
def test_lightgcn_script_is_surveyed_with_a_decay_free_optimizer_and_the_loss_side_l2(root):
    from recboard_amd import bridge
    mod = import_script(os.path.join(root, "LightGCN", "main.py"), "_bridge_lgcn_" + str(abs(hash(root))), [])
    z, model, coach = _lightgcn_case(mod, "CoachForLightGCN")
    cls, spec, plan, grads, sched = bridge.survey(coach)
    assert cls is bridge.LightGCNAdapter and sched is None
    assert set(spec.wd.values()) == {0.0}                        # LightGCN/main.py:139-145: the optimizer carries no weight decay ...
    assert plan["weight_decay"] == float(z["cfg/weight_decay"]) and plan["num_layers"] == int(z["cfg/num_layers"])   # ... the loss does (:160)
    for k in ("User.embeddings.weight", "Item.embeddings.weight"):   # golden gradients are of rec + weight_decay * emb
        np.testing.assert_allclose(grads[k].numpy(), z["grad/" + k], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("root", SCRIPTS)
def test_deepfm_script_is_surveyed_with_its_two_decay_groups_clip_and_plateau_schedule(root):
    from recboard_amd import bridge
    mod = import_script(os.path.join(root, "DeepFM", "main.py"), "_bridge_dfm_" + str(abs(hash(root))), DEEPFM_ARGS)
    z, model, coach = _deepfm_case(mod, "CoachForDeepFM")
    sched_before = dict(coach.lr_scheduler.state_dict())
    cls, spec, (kw, fields, vmap), grads, sched = bridge.survey(coach)
    assert cls is bridge.DeepFMAdapter and sched == "front_best"           # DeepFM/main.py:256: lr_scheduler.step(self._best) in front of the loop
    assert kw["embedding_decay"] == mod.cfg.embedding_decay and kw["weight_decay"] == mod.cfg.weight_decay and kw["batch_norm"] is True
    assert kw["hidden_dims"] == (32, 24, 16) and kw["hidden_dropout_rate"] == 0.3 and kw["counts"] == z["cfg/counts"].tolist()
    assert sorted(v for v in vmap.values() if isinstance(v, tuple)) == sorted([("T", f) for f in range(10)] + [("TL", f) for f in range(10)])
    # gradients handed to optimizer.step(): the golden ones after clip_grad_norm_(., 10) (dropout off in the probe, as in the golden)
    ref = {}
    for name, key in vmap.items():
        ref[name] = z[("gtable/%d" if key[0] == "T" else "gtable_lr/%d") % key[1]] if isinstance(key, tuple) else z["grad/" + key]
    total = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in ref.values()))
    clip = min(1.0, 10.0 / (total + 1e-6))
    for name, g in ref.items():
        np.testing.assert_allclose(grads[name].numpy().reshape(g.shape), g * clip, rtol=2e-4, atol=1e-7, err_msg=name)
    assert {k: v for k, v in coach.lr_scheduler.state_dict().items() if k != "_last_lr"} == {k: v for k, v in sched_before.items() if k != "_last_lr"}


def test_survey_refuses_other_optimizers_accumulation_and_per_batch_schedules():
    from recboard_amd import bridge
    mod = import_script(os.path.join(ROOT, "examples", "MF-BPR", "main.py"), "_bridge_mf_refuse", [])

    class SGDCoach(mod.CoachForMF):
        def set_optimizer(self):
            self.optimizer = torch.optim.SGD(self.model.parameters(), lr=0.1)

    class AccumCoach(mod.CoachForMF):
        def train_per_epoch(self, epoch):
            for i, data in enumerate(self.dataloader):
                self.model(self.dict_to_device(data))["rec_loss"].backward()
                if i % 2 == 1:
                    self.optimizer.step()
                    self.optimizer.zero_grad()

    class StepSchedCoach(mod.CoachForMF):
        def set_lr_scheduler(self):
            self.lr_scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, 1)

        def train_per_epoch(self, epoch):
            for data in self.dataloader:
                loss = self.model(self.dict_to_device(data))["rec_loss"]
                self.optimizer.zero_grad(); loss.backward(); self.optimizer.step(); self.lr_scheduler.step()

    class EpochSchedCoach(StepSchedCoach):
        def train_per_epoch(self, epoch):
            mod.CoachForMF.train_per_epoch(self, epoch)
            self.lr_scheduler.step()

    class AmsCoach(mod.CoachForMF):
        def set_optimizer(self):
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=1e-3, amsgrad=True)

    for cls, why in ((SGDCoach, "SGD"), (AccumCoach, "accumulation"), (StepSchedCoach, "schedule"), (AmsCoach, "amsgrad")):
        _, _, coach = _mf_case(mod, "CoachForMF")
        coach.__class__ = cls
        coach.set_optimizer(); coach.set_lr_scheduler()
        with pytest.raises(bridge.Refused, match=why):
            bridge.survey(coach)
    _, _, coach = _mf_case(mod, "CoachForMF")
    coach.__class__ = EpochSchedCoach
    coach.set_optimizer(); coach.set_lr_scheduler()
    assert bridge.survey(coach)[4] == "back"
    assert coach.lr_scheduler.last_epoch == 0               # (the probe's scheduler step was rolled back)


def test_missing_engine_library_is_reported_loudly(monkeypatch):
    """cfg.engine = "auto" on a CUDA device with no loadable librecengine.so: a warning that says so -- not a silent torch run."""
    import freerec
    from recboard_amd import lib
    mod = import_script(os.path.join(ROOT, "examples", "MF-BPR", "main.py"), "_bridge_mf_nolib", [])
    _, _, coach = _mf_case(mod, "CoachForMF")
    coach.device = torch.device("cuda:0")                    # (only the attach decision is exercised: nothing touches the device)
    monkeypatch.setattr(lib, "load", lambda: (_ for _ in ()).throw(OSError("librecengine.so: cannot open shared object file")))
    with pytest.warns(UserWarning, match="librecengine.so is not available"):
        assert freerec.launcher.Coach._attach_engine(coach) is None


def test_pointer_marshalling_notes_storages_only_while_a_recording_is_open():
    """recboard_amd/capture.py: recording() hangs every storage whose address a launch was handed on the graph (a hipGraph replays raw addresses).
    The host half, no GPU: ops._p / ops._note (what _ptr_table and adam_fuse go through) note storages into ops._KEEP while it is a list and
    not otherwise; a view notes its BASE storage (what has to stay alive), None is passed through."""
    from recboard_amd import ops
    base = torch.zeros(64)
    view = base[16:32]
    assert ops._KEEP is None
    assert ops._p(view).value == view.data_ptr() and ops._p(None) is None and ops._KEEP is None
    ops._KEEP = keep = []
    try:
        ops._p(view)
        assert ops._note(base) == base.data_ptr() and ops._note(view) == view.data_ptr()
    finally:
        ops._KEEP = None
    assert len(keep) == 3 and all(st.data_ptr() == base.untyped_storage().data_ptr() for st in keep)
    ops._p(view)
    assert len(keep) == 3
