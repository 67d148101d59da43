"""Training state -> .fearw (feartracker_amd/export.py): BatchNorm folding checked against the eval-mode forward of the
training graph (CPU), and the exported file through the HIP inference engine (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle.fear_oracle import OracleNet
from oracle.fear_train_oracle import FEARNetTrainOracle, random_init_state

from feartracker_amd.export import export_training_state, fold_training_state


def _eval_reference(sd, tmpl, srch):
    net = FEARNetTrainOracle()
    net.load_state_dict(sd, strict=False)
    net.eval()
    with torch.no_grad():
        bbox, cls = net(tmpl, srch)
    return bbox, cls


def _inputs(n=2, seed=11):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, 128, 128, generator=g), torch.randn(n, 3, 256, 256, generator=g)


def test_block_table_matches_the_shipped_model():
    """Same block kinds, roles, residual flags, activations and conv shapes as the .fearw decoded from the reference's .mlmodel."""
    import struct
    from conftest import WEIGHTS
    buf = open(WEIGHTS, "rb").read()
    _, _, n_convs, n_blocks, _, _ = struct.unpack_from("<8s4IQ", buf, 0)
    shipped_convs = [struct.unpack_from("<8I", buf, 64 + 72 * i) for i in range(n_convs)]
    off = 64 + 72 * n_convs
    shipped_blocks = [struct.unpack_from("<2I3i3I", buf, off + 32 * i) for i in range(n_blocks)]
    convs, blocks = fold_training_state(random_init_state(3))
    assert len(blocks) == n_blocks and len(convs) == n_convs

    def shape(c):
        return (c["cout"], c["cin_g"], c["groups"], c["k"], c["stride"], c["pad"], c["relu"], int(c["b"] is not None))
    for mine, ref in zip(blocks, shipped_blocks):
        kind, role, c0, c1, c2, residual, act, _ = ref
        assert (mine["kind"], mine["role"], mine["residual"], mine["act"]) == (kind, role, residual, act)
        for a, b in zip(mine["conv"], (c0, c1, c2)):
            assert (a < 0) == (b < 0)
            if a >= 0:
                assert shape(convs[a]) == tuple(shipped_convs[b])


@pytest.mark.parametrize("payload,tol", [("fp32", 2e-5), ("fp16", 2e-2)])
def test_folded_weights_reproduce_the_eval_forward(tmp_path, payload, tol):
    """eval-mode forward of the training graph (BatchNorm on running statistics, exp(adjust * x + bias), 0.1 * cls) ==
    the inference restatement on the exported file; fp16 payload within half-precision rounding of 67 weight tensors."""
    sd = random_init_state(3)
    path = os.path.join(tmp_path, "exported.fearw")
    export_training_state(sd, path, payload=payload)
    tmpl, srch = _inputs()
    bbox, cls = _eval_reference(sd, tmpl, srch)
    ora = OracleNet(path)
    out = ora.track(srch, ora.get_features(tmpl))
    eb = float((out["TARGET_REGRESSION_LABEL_KEY"] - bbox).abs().max() / bbox.abs().max())
    ec = float((out["TARGET_CLASSIFICATION_KEY"] - cls).abs().max() / cls.abs().max())
    assert eb < tol and ec < tol, (eb, ec)


@pytest.mark.gpu
def test_exported_state_runs_on_the_hip_engine(tmp_path):
    """train state -> export (fp32 payload) -> FEARNetHIP: the engine's maps match the eval-mode training graph to 1e-3
    (north_star tolerance), on the throughput plan and on the one-crop plan."""
    from feartracker_amd import FEARNetHIP
    sd = random_init_state(3)
    path = os.path.join(tmp_path, "exported.fearw")
    export_training_state(sd, path, payload="fp32")
    tmpl, srch = _inputs(n=3)
    bbox, cls = _eval_reference(sd, tmpl, srch)
    for max_batch in (256, 1):
        net = FEARNetHIP(path, device=0, max_batch=max_batch)
        b, c = net.track_maps(srch.cuda(), net.get_features(tmpl.cuda()))
        eb = float((b.cpu() - bbox).abs().max() / bbox.abs().max())
        ec = float((c.cpu() - cls).abs().max() / cls.abs().max())
        assert eb < 1e-3 and ec < 1e-3, (max_batch, eb, ec)
