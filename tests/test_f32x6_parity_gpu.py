"""GPU: the core fp32 parity tests once more with `f32x6` — fp32 emulated on the bf16 matrix pipe in the forward / data-gradient GEMMs (every operand split exactly into
three bf16 pieces, six products, sign-alternating fp32 accumulation) — as the process default, so that the driver's GPU run, not a builder log, shows them
green.  The mode stays a labelled secondary key of bench.py; the headline and every parity claim are native fp32."""
import pytest
import torch

from tests import test_b70_gpu as T70
from tests import test_fullwidth_gpu as TFW
from tests import test_models_gpu as TM

pytestmark = pytest.mark.gpu


@pytest.fixture()
def x6():
    from dcvgan_amd import native
    native.lib()
    native.set_precision("f32x6")
    try:
        yield torch.device("cuda:0")
    finally:
        native.set_precision("fp32")


@pytest.mark.parametrize("elide", [False, True], ids=["as_written", "dead_backward_elided"])
@pytest.mark.parametrize("fixture", ["step_depth_adv_g1.npz", "step_depth_adv_g1_evalstart.npz", "step_flow_hinge_g2.npz"])
def test_training_step_f32x6(x6, fixture, elide):
    TM.test_training_step(x6, fixture, elide)


@pytest.mark.parametrize("fixture", ["step_fullwidth_isogd_depth.npz", "step_fullwidth_surreal_depth1.npz", "step_fullwidth_isogd_flow.npz"])
def test_fullwidth_training_step_f32x6(x6, fixture):
    TFW.test_fullwidth_training_step(x6, fixture)


def test_batch_split_identity_f32x6(x6):
    T70.test_batch_split_identity(x6)


def test_the_mode_really_ran_on_the_bf16_pipe(x6):
    from dcvgan_amd import native, ops
    x = torch.randn(8, 128, 32, 32, device=x6); w = torch.randn(128, 64, 4, 4, device=x6) * 0.05
    with torch.no_grad():
        ops.conv(x, w, ops.conv_geom(w, (2, 2), (1, 1), True))
    kn = native.lib().dcv_debug_last_kernel().decode()
    assert "x6" in kn.lower() or "f32x6" in kn.lower() or "3 x bf16" in kn.lower(), kn
