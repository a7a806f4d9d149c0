"""MI355X-native AES-GCM bulk path behind the software-model surface of
BLu85/AES-GCM-128-192-256-bits (tb/gcm_model.py).

    import aesgcm_amd                      # registers this directory under an importable name
    from aesgcm_amd import gcm_model       # class gcm(key, icb, ed) + encrypt()/decrypt()
    from aesgcm_amd import lib             # thin ctypes binding of libaesgcm_hip.so (include/aesgcm.h)

Python here is host plumbing only: every arithmetic step runs in the HIP kernels of csrc/.  There
is no CPU fallback: if the shared library or a GPU is missing the calls raise.
"""
from . import lib            # noqa: F401
from . import gcm_model      # noqa: F401
from . import sharding       # noqa: F401
from . import comm           # noqa: F401
from .gcm_model import gcm, encrypt, decrypt, AesGcmError, AuthenticationError   # noqa: F401
from .build import build     # noqa: F401
