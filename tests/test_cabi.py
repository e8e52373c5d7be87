"""CPU: the C-ABI library loads and exports every symbol include/wcn.h declares (no compute calls)."""
import ctypes
import os
import re

from tests.conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "wcn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wcn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(hip_lib):
    from warpconvnet_amd import _lib

    declared = _declared_symbols()
    assert len(declared) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(raw, s)]
    assert not missing, f"declared in wcn.h but not exported: {missing}"
    # the ctypes table binds exactly the declared entry points
    assert sorted(_lib.SIGNATURES) == declared


def test_pure_host_entry_points(hip_lib):
    """Entry points that do not touch the device."""
    from warpconvnet_amd import _lib

    L = hip_lib
    assert L.wcn_abi_version() >= 1
    assert _lib.status_string(0) == "Success" and _lib.status_string(-4) == "Unsupported precision/configuration"
    assert _lib.status_string(-5) == "Invalid parameters" and _lib.status_string(-99) == "Unknown error"
    assert L.wcn_kmap_row_pitch(27) == 32 and L.wcn_kmap_row_pitch(8) == 8 and L.wcn_kmap_row_pitch(125) == 128
    assert L.wcn_kmap_mask_words(27) == 1 and L.wcn_kmap_mask_words(33) == 2
    assert L.wcn_kmap_num_blocks(1000) == 4 and L.wcn_kmap_num_blocks(1025) == 8
    assert L.wcn_kmap_counts_bytes(1000, 27) >= (27 * 5 + 1) * 4
    assert L.wcn_mfma_gather_supported(64, 128, 27, _lib.WCN_BF16) == 1
    assert L.wcn_mfma_gather_supported(64, 128, 27, _lib.WCN_F32) == 0
    assert L.wcn_mfma_gather_supported(7, 13, 27, _lib.WCN_BF16) == 0
    assert L.wcn_mfma_gather_supported(64, 128, 125, _lib.WCN_BF16) == 1  # multi-word masks
    assert L.wcn_mfma_gather_supported(64, 128, 2000, _lib.WCN_BF16) == 0
    assert L.wcn_mfma_wgrad_supported(64, 128, _lib.WCN_F16) == 1 and L.wcn_mfma_wgrad_supported(96, 32, _lib.WCN_F16) == 1
    assert L.wcn_mfma_wgrad_supported(48, 64, _lib.WCN_F16) == 0 and L.wcn_mfma_wgrad_supported(64, 64, _lib.WCN_F32) == 0
    assert L.wcn_mfma_gather_supported(64, 192, 27, _lib.WCN_BF16) == 1
    assert L.wcn_kmap_binned_supported(_lib.i3((3, 3, 3)), _lib.i3((1, 1, 1))) == 1
    assert L.wcn_kmap_binned_supported(_lib.i3((3, 3, 3)), _lib.i3((8, 8, 8))) == 1  # halo 8 = one block width
    assert L.wcn_kmap_binned_supported(_lib.i3((3, 3, 3)), _lib.i3((9, 1, 1))) == 0  # beyond: hash path
    assert L.wcn_pointconv_supported(32, 32, 0, 128, 64, 16, 0) == 1 and L.wcn_pointconv_supported(24, 24, 0, 128, 64, 16, 0) == 0
    assert L.wcn_pointconv_supported(24, 24, 0, 128, 64, 16, 1) == 1 and L.wcn_pointconv_supported(32, 32, 0, 128, 64, 12, 0) == 0
    assert L.wcn_pointconv_grad_floats(64, 128, 64, 1) - L.wcn_pointconv_grad_floats(64, 128, 64, 0) == 64 * 64 + 64
    assert L.wcn_packed_weight_bytes(27, 64, 128, _lib.WCN_BF16, 0) == 27 * 64 * 128 * 2
    assert L.wcn_packed_weight_bytes(27, 96, 96, _lib.WCN_BF16, 0) == 27 * 128 * 96 * 2  # trailing 32-channel chunk zero-padded
    assert L.wcn_conv_wgrad_workspace(27, 64, 128, _lib.WCN_ALGO_MFMA) > 27 * 64 * 128 * 4
    assert L.wcn_mask_argsort_workspace(1000) > 3 * 4000 and L.wcn_kmap_binned_workspace(1000, 250) > 250 * 2048
    assert L.wcn_kmap_binned_supported(_lib.i3((4, 4, 2)), _lib.i3((1, 1, 1))) == 0  # K % 32 == 0: hash path
    assert L.wcn_kmap_tally_sort_workspace(1000) >= L.wcn_mask_argsort_workspace(1000)
    # the packing entry points check the destination size (ABI 2): a short buffer is refused before any launch
    import ctypes as _ct
    buf = (_ct.c_char * 64)()
    assert L.wcn_abi_version() >= 2
    assert L.wcn_pack_weight(_ct.addressof(buf), 27, 64, 128, _lib.WCN_BF16, 0, 0, _ct.addressof(buf), 64, None) == -5
    assert L.wcn_pack_weight_f32(_ct.addressof(buf), 27, 96, 96, _lib.WCN_BF16, 0, 0, _ct.addressof(buf), 27 * 96 * 96 * 2, None) == -5
    # ABI 3 additions: identity-map dense products, residual-tail BatchNorm passes
    assert L.wcn_abi_version() >= 3
    assert L.wcn_conv_identity_supported(96, 128, _lib.WCN_BF16) == 1 and L.wcn_conv_identity_supported(128, 256, _lib.WCN_F16) == 1
    # every shape the identity-map / channel-split path announces is one wcn_conv_gather_gemm accepts (320 = 5 x 64 was not)
    for cin, cout in ((64, 320), (128, 384), (96, 512), (256, 192)):
        assert L.wcn_conv_identity_supported(cin, cout, _lib.WCN_BF16) == 1
        assert L.wcn_mfma_gather_supported(cin, cout, 1, _lib.WCN_BF16) == 1 and L.wcn_mfma_gather_supported(cin, cout, 27, _lib.WCN_F16) == 1
    assert L.wcn_conv_identity_supported(48, 64, _lib.WCN_BF16) == 0 and L.wcn_conv_identity_supported(96, 20, _lib.WCN_BF16) == 0
    assert L.wcn_conv_identity_supported(64, 64, _lib.WCN_F32) == 0
    assert L.wcn_bn_apply_residual(None, None, 4, 8, _lib.WCN_BF16, None, None, 1, None, None) == -5
    # ABI 4 additions: tile order as an entry point, both weight images in one launch, the row mask in the table's last column
    assert L.wcn_abi_version() >= 4
    # ABI 5: compact neighbour rows (17 <= K <= 31) replace the mask-in-column-31 convention of ABI 4
    assert L.wcn_abi_version() >= 5
    assert L.wcn_kmap_compact_supported(27) == 1 and L.wcn_kmap_compact_supported(25) == 1
    assert L.wcn_kmap_compact_supported(9) == 0 and L.wcn_kmap_compact_supported(32) == 0 and L.wcn_kmap_compact_supported(125) == 0
    assert L.wcn_conv_compact_table_supported(64, 128, 27, _lib.WCN_BF16) == 1
    assert L.wcn_conv_compact_table_supported(64, 128, 32, _lib.WCN_BF16) == 0  # two mask words
    assert L.wcn_conv_compact_table_supported(32, 32, 27, _lib.WCN_BF16) == 0   # not a channel-split shape
    assert L.wcn_kmap_densify(None, 4, 27, None, None) == -5 and L.wcn_kmap_densify(None, 0, 27, None, None) == 0
    assert L.wcn_kmap_densify(None, 4, 9, None, None) == -5
    assert L.wcn_pack_weight_pair_supported(27, 64, 128, _lib.WCN_BF16) == 1
    # narrow 1 x 1 x 1 layers (stem / head) as one streaming launch
    assert L.wcn_dense_rows_supported(96, 20, _lib.WCN_BF16) == 1 and L.wcn_dense_rows_supported(3, 32, _lib.WCN_F16) == 1
    assert L.wcn_dense_rows_supported(129, 20, _lib.WCN_BF16) == 0 and L.wcn_dense_rows_supported(96, 20, _lib.WCN_F32) == 0
    assert L.wcn_dense_rows(None, 0, None, 1, 0, None, None, 4, 96, 20, _lib.WCN_BF16, None) == -5
    assert L.wcn_dense_rows(None, 0, None, 1, 0, None, None, 4, 96, 200, _lib.WCN_BF16, None) != 0
    assert L.wcn_bn_backward_reduce_masked(None, None, None, 4, 8, _lib.WCN_BF16, None, None, None, None, None, 0, None) == -5
    assert L.wcn_bn_backward_apply_masked(None, None, None, 4, 8, _lib.WCN_BF16, None, None, None, None, None, None, None, None) == -5
    assert L.wcn_bn_train_forward(None, None, 4, 8, _lib.WCN_BF16, None, None, None, None, 0.1, 1e-5, None, 1, None, None, None, 0, None) == -5
    assert L.wcn_bn_train_backward(None, None, None, 1, 4, 8, _lib.WCN_BF16, None, None, 1, None, None, None, None, 0, None) == -5
    assert L.wcn_conv_bn_backward(*([None] * 4), 1, None, None, 1, *([None] * 7), 0, *([None] * 6), 0, 4, 4, 64, 64, 27, _lib.WCN_BF16,
                                  None, 0, None) == -5
    # parameter validation happens before any launch: bad arguments come back as status codes
    assert L.wcn_hash_prepare(None, 16, None) == -5
    assert L.wcn_hash_prepare(None, 17, None) == -5
    assert L.wcn_conv_gather_gemm(None, None, None, None, None, None, None, 0, 0, 0, 0, 0, 0, 1, 0, 0, None) == -5


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from warpconvnet_amd import _lib

    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.lib()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("lib() must raise when the HIP extension is missing")
