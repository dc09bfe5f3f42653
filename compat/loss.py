"""Drop-in alias: put this directory on sys.path and `import loss` resolves to the HIP-backed
implementation with the reference's names (src/loss.py).  See INTEGRATION.md."""
from dcvgan_amd.loss import *  # noqa: F401,F403
from dcvgan_amd import loss as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
