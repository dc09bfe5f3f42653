"""Drop-in alias: put this directory on sys.path and `import util` resolves to the HIP-backed
implementation with the reference's names (src/util.py).  See INTEGRATION.md."""
from dcvgan_amd.util import *  # noqa: F401,F403
from dcvgan_amd import util as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
