"""nowcastautogp_amd — MI355X-native GP inference core behind the AutoGP surface that
NowcastAutoGP's ``make_and_fit_model`` / ``forecast`` / ``forecast_with_nowcasts`` call.

The compute path is ``libngp.so`` (hand-written HIP for gfx950, C-ABI in ``include/ngp.h``);
this package is the thin host-side mirror of the reference interface above it.
"""
from . import gp  # noqa: F401
from ._abi import default_spec  # noqa: F401

__version__ = "0.1.0"
