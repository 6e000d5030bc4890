"""MI355X-native encrypted-string engine (C ABI in include/fhe_string_amd.h)."""
from ._lib import Params, Stats, FheError, load  # noqa: F401
