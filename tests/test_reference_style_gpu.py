"""GPU: the reference's own module tests (src/test/test_generator.py:18-51, src/test/test_discriminator.py:14-76), restated on
the HIP-backed classes with the same keyword constructors, default widths, inputs and expected shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
IMAGE_SIZE, VIDEO_LENGTH, BATCHSIZE, COLOR_CH, GEOMTRIC_INFO_CH = 64, 16, 2, 3, 1
geometric_infos = {"depth": 1, "optical-flow": 2}


@pytest.fixture(scope="module")
def dev():
    from dcvgan_amd import native
    native.lib()
    return torch.device("cuda:0")


def test_depth_video_generator(dev):
    from dcvgan_amd.generator import GeometricVideoGenerator
    for name, n_channels in geometric_infos.items():
        ggen = GeometricVideoGenerator(dim_z_content=30, dim_z_motion=10, channel=n_channels, geometric_info=name, video_length=VIDEO_LENGTH).to(dev)
        videos = ggen.sample_videos(BATCHSIZE)
        assert videos.shape == (BATCHSIZE, n_channels, VIDEO_LENGTH, IMAGE_SIZE, IMAGE_SIZE)
        assert float(videos.abs().max()) <= 1.0 and bool(torch.isfinite(videos).all())     # Tanh head


def test_color_video_generator(dev):
    from dcvgan_amd.generator import ColorVideoGenerator
    for name, n_channels in geometric_infos.items():
        cgen = ColorVideoGenerator(in_ch=n_channels, dim_z=10, geometric_info=name).to(dev)
        x = torch.empty((BATCHSIZE, n_channels, IMAGE_SIZE, IMAGE_SIZE), device=cgen.device).normal_()
        z = cgen.make_hidden(BATCHSIZE)
        output = cgen(x, z)
        assert output.shape == (BATCHSIZE, COLOR_CH, IMAGE_SIZE, IMAGE_SIZE)
        xs = torch.empty((BATCHSIZE, n_channels, VIDEO_LENGTH, IMAGE_SIZE, IMAGE_SIZE), device=cgen.device).normal_()
        assert cgen.forward_videos(xs).shape == (BATCHSIZE, COLOR_CH, VIDEO_LENGTH, IMAGE_SIZE, IMAGE_SIZE)


@pytest.mark.parametrize("cls,expected,video", [("ImageDiscriminator", (BATCHSIZE, 4, 4), False), ("VideoDiscriminator", (BATCHSIZE, 4, 4, 4), True),
                                                ("GradientDiscriminator", (BATCHSIZE, 3, 4, 4), True)])
def test_discriminators(dev, cls, expected, video):
    from dcvgan_amd import discriminator
    dis = getattr(discriminator, cls)(ch1=GEOMTRIC_INFO_CH, ch2=COLOR_CH, use_noise=True, noise_sigma=0.2).to(dev)
    t = (VIDEO_LENGTH,) if video else ()
    xg = torch.empty((BATCHSIZE, GEOMTRIC_INFO_CH) + t + (IMAGE_SIZE, IMAGE_SIZE), device=dis.device).normal_()
    xc = torch.empty((BATCHSIZE, COLOR_CH) + t + (IMAGE_SIZE, IMAGE_SIZE), device=dis.device).normal_()
    output = dis(xg, xc)
    assert output.shape == expected
    # B = 1: `.squeeze()` also drops the batch dimension, as in the reference (discriminator.py:127,231,333)
    assert dis(xg[:1], xc[:1]).shape == expected[1:]
