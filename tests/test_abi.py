"""The C-ABI library loads and exports every symbol include/fear_hip.h declares; error paths that
need no GPU behave (no compute calls here)."""
import ctypes
import os
import re

import pytest

from feartracker_amd import hip_backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fear_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fear_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = hip_backend.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 14
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/fear_hip.h but not exported"
    assert set(declared) == set(hip_backend.EXPORTED_SYMBOLS)
    assert b"gfx950" in lib.fear_version()


def test_status_strings_and_null_handling():
    lib = hip_backend.load_library()
    assert lib.fear_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -6):
        assert lib.fear_strerror(code) not in (b"ok", b"unknown status")
    assert lib.fear_strerror(-99) == b"unknown status"
    h = ctypes.c_void_p()
    assert lib.fear_create(None, 0, 0, ctypes.byref(h)) == -1
    assert lib.fear_create(b"not a model" * 20, 220, 0, ctypes.byref(h)) == -3        # FEAR_ERR_FORMAT
    assert lib.fear_destroy(None) == -1
    assert lib.fear_set_option(None, 1, 8) == -1
    assert lib.fear_workspace_bytes(None) == 0


def test_option_ids_match_header():
    text = open(os.path.join(ROOT, "include", "fear_hip.h")).read()
    ids = dict(re.findall(r"#define (FEAR_OPT_[A-Z_0-9]+) (\d+)", text))
    assert ids == {"FEAR_OPT_MAX_BATCH": "1", "FEAR_OPT_PROFILE": "2", "FEAR_OPT_PROFILE_OP": "3", "FEAR_OPT_FUSE": "4",
                   "FEAR_OPT_MATH": "5", "FEAR_OPT_CHAIN": "6", "FEAR_OPT_SMALL_PASS": "7", "FEAR_OPT_PLAN_CROPS": "8",
                   "FEAR_OPT_DUAL_HEAD": "9", "FEAR_OPT_HEAD_STAGGER": "10", "FEAR_OPT_TILE_V4": "11", "FEAR_OPT_TINY_SEP": "12",
                   "FEAR_OPT_HEAD_CHAIN": "13", "FEAR_OPT_BF16_STORE": "14", "FEAR_OPT_E1_PAIR": "15", "FEAR_OPT_SPLIT_STREAMS": "16",
                   "FEAR_OPT_CHAIN32": "17"}
    for name, val in ids.items():
        assert getattr(hip_backend, name) == int(val)


def test_truncated_model_is_rejected():
    lib = hip_backend.load_library()
    blob = open(hip_backend.DEFAULT_WEIGHTS, "rb").read()
    h = ctypes.c_void_p()
    assert lib.fear_create(blob[: len(blob) // 2], len(blob) // 2, 0, ctypes.byref(h)) == -3
    bad = bytearray(blob)
    bad[8] = 9          # version field
    assert lib.fear_create(bytes(bad), len(bad), 0, ctypes.byref(h)) == -3


def test_malformed_tables_are_rejected_not_followed():
    """conv indices below -1 / missing mandatory convs, offsets that wrap, unbounded dimensions must give
    FEAR_ERR_FORMAT (parse happens before any device call, so this runs without a GPU)."""
    import struct
    lib = hip_backend.load_library()
    blob = open(hip_backend.DEFAULT_WEIGHTS, "rb").read()
    magic, version, n_convs, n_blocks, dtype, payload = struct.unpack_from("<8s4IQ", blob, 0)
    conv0, blk0 = 64, 64 + 72 * n_convs
    h = ctypes.c_void_p()

    def create(b):
        return lib.fear_create(bytes(b), len(b), 0, ctypes.byref(h))

    bad = bytearray(blob)
    struct.pack_into("<i", bad, blk0 + 32 * 1 + 8 + 4, -7)            # block 1 (an IR block): depthwise index -7
    assert create(bad) == -3
    bad = bytearray(blob)
    struct.pack_into("<i", bad, blk0 + 32 * 1 + 8 + 8, -1)            # ... mandatory project conv missing
    assert create(bad) == -3
    bad = bytearray(blob)
    struct.pack_into("<i", bad, blk0 + 8, -1)                         # stem conv missing
    assert create(bad) == -3
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, conv0 + 32, 2 ** 64 - 16)             # w_off that wraps when the size is added
    assert create(bad) == -3
    bad = bytearray(blob)
    struct.pack_into("<I", bad, conv0, 2 ** 31)                       # cout far beyond any model
    assert create(bad) == -3
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, 24, 2 ** 64 - 8)                      # payload_bytes that wraps the size check
    assert create(bad) == -3
    bad = bytearray(blob)
    struct.pack_into("<I", bad, conv0 + 12, 7)                        # kernel size 7
    assert create(bad) == -3
    bad = bytearray(blob)
    struct.pack_into("<I", bad, 20, 2)                                # payload_dtype: only 0 (fp16) and 1 (fp32) exist
    assert create(bad) == -3
    bad = bytearray(blob)
    struct.pack_into("<I", bad, 20, 1)                                # an fp16 payload declared fp32: the last convs run past it
    assert create(bad) == -3


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        hip_backend.FEARNetHIP()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "feartracker_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.replace("no oracle", ""), f"{fn} mentions the oracle"


@pytest.mark.gpu
def test_error_codes_and_options_on_device():
    """Status codes of the entry points on a live handle: empty batches are fine with null tensors, negative sizes and
    null tensors are reported (never a crash), options round-trip and reject out-of-range values."""
    import ctypes
    import torch
    from feartracker_amd import FEARNetHIP
    from feartracker_amd import hip_backend as hb
    from conftest import WEIGHTS
    net = FEARNetHIP(WEIGHTS, device=0, max_batch=4)
    lib, h = net._lib, net._h
    OK, NULL, SHAPE = 0, -1, -2
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.fear_track(h, None, None, None, 0, None, None, st) == OK
    assert lib.fear_track(h, None, None, None, -1, None, None, st) == SHAPE
    assert lib.fear_track(h, None, None, None, 2, None, None, st) == NULL
    assert lib.fear_track_packed(h, None, None, None, 0, None, st) == OK
    assert lib.fear_track_packed(h, None, None, None, 2, None, st) == NULL
    assert lib.fear_features(h, None, 0, 128, None, st) == OK
    assert lib.fear_features(h, None, 1, 128, None, st) == NULL
    x = torch.zeros(1, 3, 100, 100, device="cuda")
    out = torch.zeros(1, 256, 8, 8, device="cuda")
    assert lib.fear_features(h, x.data_ptr(), 1, 100, out.data_ptr(), st) == SHAPE       # not a multiple of 32
    assert lib.fear_decode(h, None, None, 0, 16, 16, 256, None, None, None, st) == OK
    assert lib.fear_decode(h, None, None, 3, 16, 16, 256, None, None, None, st) == NULL
    assert lib.fear_decode(h, None, None, 3, 0, 16, 256, None, None, None, st) == SHAPE
    assert lib.fear_decode_smooth(h, None, None, 0, 16, 16, 256, None, None, 0.1, 0.3, 0.3, None, None, None, st) == OK
    assert lib.fear_decode_smooth(h, None, None, 2, 16, 16, 256, None, None, 0.1, 0.3, 0.3, None, None, None, st) == NULL
    for opt, good, bad in ((hb.FEAR_OPT_MAX_BATCH, 17, 0), (hb.FEAR_OPT_MATH, 2, 3), (hb.FEAR_OPT_CHAIN, 0, 5),
                           (hb.FEAR_OPT_SMALL_PASS, 12, -1), (hb.FEAR_OPT_FUSE, 0, 3), (hb.FEAR_OPT_PLAN_CROPS, 3, -2), (hb.FEAR_OPT_DUAL_HEAD, 1, 2)):
        assert lib.fear_set_option(h, opt, good) == OK and lib.fear_get_option(h, opt) == good
        assert lib.fear_set_option(h, opt, bad) == SHAPE and lib.fear_get_option(h, opt) == good
    assert lib.fear_set_option(h, 999, 1) == SHAPE
