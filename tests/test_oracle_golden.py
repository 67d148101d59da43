"""Pin the CPU oracle (oracle/fear_oracle.py) against fixtures produced by the real reference
(tools/make_golden.py): literal CoreML-graph execution for the trunk, the reference's own
AdjustLayer/BoxTower/FEARBoxCoder classes for neck, head and decode."""
import numpy as np
import torch

from oracle.fear_oracle import OracleNet, decode, make_grid, normalize_u8


def _norm(u8_nchw):
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1) * 255.0
    inv = 1.0 / (torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1) * 255.0)
    return (u8_nchw.float() - mean) * inv


def test_whole_net_maps_match_coreml_graph(oracle_net, golden_dir):
    d = np.load(f"{golden_dir}/track_maps.npz")
    x = _norm(torch.from_numpy(d["search_u8"][:3]))
    z = torch.from_numpy(d["template_features"][:3])
    out = oracle_net.track(x, z)
    np.testing.assert_allclose(out["TARGET_REGRESSION_LABEL_KEY"].numpy(), d["bbox"][:3], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out["TARGET_CLASSIFICATION_KEY"].numpy(), d["cls"][:3], rtol=1e-5, atol=1e-5)
    zt = oracle_net.get_features(_norm(torch.from_numpy(d["template_u8"][:3])))
    np.testing.assert_allclose(zt.numpy(), d["template_features"][:3], rtol=1e-5, atol=1e-5)


def test_trunk_block_taps(oracle_net, golden_dir):
    d = np.load(f"{golden_dir}/trunk_taps.npz")
    taps = []
    oracle_net.feature_extractor(torch.from_numpy(d["image"]), taps)
    assert len(taps) == 17
    for i, t in enumerate(taps):
        np.testing.assert_allclose(t.numpy(), d[f"block{i:02d}"], rtol=1e-5, atol=1e-5)


def test_neck_and_head_match_reference_modules(oracle_net, golden_dir):
    """Fixture = reference AdjustLayer + BoxTower (model/blocks.py:75-194) run with the CoreML weights."""
    d = np.load(f"{golden_dir}/head_modules.npz")
    neck = oracle_net.neck[0]["conv"][0]
    xs = oracle_net._conv(neck, torch.from_numpy(d["trunk_out"]))
    zs = oracle_net._conv(neck, torch.from_numpy(d["tmpl_trunk"]))
    zu = oracle_net._conv(neck, torch.from_numpy(d["upd_trunk"]))
    np.testing.assert_allclose(xs.numpy(), d["neck_search"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(zs.numpy(), d["neck_template"], rtol=1e-4, atol=1e-4)
    bbox, cls, cls_dw, x_reg = oracle_net.connector(zs, xs, return_all=True)

    def close(a, b, rel=2e-5):
        assert float(np.abs(a - b).max()) <= rel * float(np.abs(b).max()), (np.abs(a - b).max(), np.abs(b).max())

    close(bbox.numpy(), d["bbox"])
    close(cls.numpy(), d["cls"])
    close(cls_dw.numpy(), d["cls_dw"])
    close(x_reg.numpy(), d["x_reg"])
    # dual-template path: `update` replaces the template of the cls branch only (blocks.py:174-179)
    bu, cu, _, _ = oracle_net.connector(zs, xs, update=zu, return_all=True)
    close(bu.numpy(), d["bbox_update"])
    close(cu.numpy(), d["cls_update"])
    close(bu.numpy(), d["bbox"])           # reg branch untouched by `update`
    assert np.abs(cu.numpy() - d["cls"]).max() > 1e-3


def test_decode_and_grid(golden_dir):
    g = np.load(f"{golden_dir}/grid_window.npz")
    gx, gy = make_grid(16, 16, 256)
    np.testing.assert_array_equal(gx, g["grid_x"])
    np.testing.assert_array_equal(gy, g["grid_y"])
    assert gx.dtype == np.float64 and list(gx[0, 0, :3]) == [0.0, 16.0, 32.0]
    d = np.load(f"{golden_dir}/box_coder.npz")
    box, rc = decode(d["reg_maps"], d["cls_maps"], use_sigmoid=True)
    np.testing.assert_array_equal(np.array(rc), d["dec_sigmoid_rc"])
    np.testing.assert_allclose(box, d["dec_sigmoid_bbox"], rtol=0, atol=1e-9)
    box, rc = decode(d["reg_maps"], d["cls_maps"], use_sigmoid=False)
    np.testing.assert_array_equal(np.array(rc), d["dec_plain_rc"])
    assert tuple(rc[3]) == (5, 7)   # exact tie -> first maximum


def test_normalize_matches_formula():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(8, 9, 3)).astype(np.uint8)
    got = normalize_u8(img)[0].numpy()
    mean = np.array([0.485, 0.456, 0.406]) * 255
    std = np.array([0.229, 0.224, 0.225]) * 255
    ref = ((img.astype(np.float64) - mean) / std).transpose(2, 0, 1)
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)
