"""Drop-in alias: put this directory on sys.path and `import discriminator` resolves to the HIP-backed
implementation with the reference's names (src/discriminator.py).  See INTEGRATION.md."""
from dcvgan_amd.discriminator import *  # noqa: F401,F403
from dcvgan_amd import discriminator as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
