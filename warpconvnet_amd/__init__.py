"""MI355X-native sparse-3D-convolution engine with the WarpConvNet module / geometry API.

Hot path: kernel-map construction + AB / ABt / AtB sparse GEMMs as hand-written HIP for gfx950,
reached through the C-ABI in ``include/wcn.h`` (``warpconvnet_amd/csrc/libwcn_hip.so``).
"""
__version__ = "0.1.0"


def _register_compile_support() -> None:
    from warpconvnet_amd import _compile

    _compile.register()


_register_compile_support()
