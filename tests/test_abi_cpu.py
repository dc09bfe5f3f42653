"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/dcvgan_hip.h
declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def lib():
    from dcvgan_amd import native
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return native.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dcvgan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dcv_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    from dcvgan_amd import native
    names = declared_symbols()
    assert len(names) >= 20
    raw = ctypes.CDLL(native.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in dcvgan_hip.h but not exported"
    assert set(names) == set(native.EXPORTS), (set(names) ^ set(native.EXPORTS))


def test_integration_md_declares_the_structs_the_library_has(lib):
    """INTEGRATION.md's binding snippet is what a maintainer copies: its struct declarations must be the ones the library was compiled
    with (round 3 shipped a 12-field dcv_conv_geom there after the 13th field was added)."""
    from dcvgan_amd import native
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class ConvGeom\(C\.Structure\):.*?_fields_ = \[\(n, C\.c_int32\) for n in \(([^)]*)\)\]", text, flags=re.S)
    assert m, "INTEGRATION.md no longer shows the ConvGeom binding"
    doc_fields = [f.strip().strip('"') for f in m.group(1).split(",")]
    assert doc_fields == [f for f, _ in native.ConvGeom._fields_]
    m = re.search(r"class WPack\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n", text, flags=re.S)
    assert m and re.findall(r'\("(\w+)"', m.group(1)) == [f for f, _ in native.WPack._fields_]
    # ... and the header's own field lists
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dcvgan_hip.h")).read(), flags=re.S)
    body = re.search(r"typedef struct dcv_conv_geom \{(.*?)\} dcv_conv_geom;", hdr, flags=re.S).group(1)
    names = [n.strip() for decl in re.findall(r"int32_t ([^;]*);", body) for n in decl.split(",")]
    assert names == [f for f, _ in native.ConvGeom._fields_]
    body = re.search(r"typedef struct dcv_wpack \{(.*?)\} dcv_wpack;", hdr, flags=re.S).group(1)
    assert re.findall(r"(\w+);", body) == [f for f, _ in native.WPack._fields_]
    sizes = (ctypes.c_size_t * 3)()
    lib.dcv_abi_struct_sizes(sizes)
    assert tuple(sizes) == (ctypes.sizeof(native.Dims5), ctypes.sizeof(native.ConvGeom), ctypes.sizeof(native.WPack))
    assert lib.dcv_version() == native.ABI_VERSION == 4
    assert re.search(r"lib\.dcv_version\(\) == 4", text)


def test_version_and_error_channel(lib):
    assert lib.dcv_version() >= 3
    assert isinstance(lib.dcv_last_error(), bytes)
    assert lib.dcv_launch_count() == 0  # nothing launched on a CPU-only box


def test_argument_validation_needs_no_gpu(lib):
    """Geometry checks run on the host before any launch: bad shapes fail with DCV_EINVAL."""
    from dcvgan_amd.native import ConvGeom, Dims5
    g = ConvGeom(1, 4, 4, 1, 2, 2, 0, 1, 1, 0, 3, 8)
    x = Dims5(2, 3, 1, 16, 16, 768, 256, 0, 16, 1)
    y_bad = Dims5(2, 8, 1, 9, 8, 576, 72, 0, 8, 1)
    y_ok = Dims5(2, 8, 1, 8, 8, 512, 64, 0, 8, 1)
    assert lib.dcv_conv_workspace_bytes(ctypes.byref(g), ctypes.byref(x), ctypes.byref(y_bad), 0) == 0
    assert b"extent" in lib.dcv_last_error()
    for which in (0, 1, 2):
        assert lib.dcv_conv_workspace_bytes(ctypes.byref(g), ctypes.byref(x), ctypes.byref(y_ok), which) > 0
    assert lib.dcv_conv_forward(ctypes.byref(g), None, ctypes.byref(x), None, None, ctypes.byref(y_ok), 0, 0.0, None, None, 0, None) == -1
    # packed-weight buffer sizes (dcv_wpack) are host arithmetic too: K = 3 channels x 16 taps = 48 rows x 32 padded output channels
    assert lib.dcv_conv_packed_bytes(ctypes.byref(g), ctypes.byref(x), ctypes.byref(y_ok), 0) == 48 * 32 * 4
    assert lib.dcv_conv_packed_bytes(ctypes.byref(g), ctypes.byref(x), ctypes.byref(y_ok), 1) > 0
    assert lib.dcv_conv_packed_bytes(ctypes.byref(g), ctypes.byref(x), ctypes.byref(y_ok), 2) == 0
    # the per-module precision field: 0 (process default), 1 (fp32), 2 (bf16 products); anything else is refused on the host
    assert ctypes.sizeof(ConvGeom) == 13 * 4 and [f for f, _ in ConvGeom._fields_][-1] == "mfma"
    g_bad = ConvGeom(1, 4, 4, 1, 2, 2, 0, 1, 1, 0, 3, 8, 4)
    assert lib.dcv_conv_workspace_bytes(ctypes.byref(g_bad), ctypes.byref(x), ctypes.byref(y_ok), 0) == 0 and b"mfma" in lib.dcv_last_error()
    g_bf = ConvGeom(1, 4, 4, 1, 2, 2, 0, 1, 1, 0, 3, 8, 2)
    assert lib.dcv_conv_workspace_bytes(ctypes.byref(g_bf), ctypes.byref(x), ctypes.byref(y_ok), 0) > 0
    # a caller-owned packed copy is stamped with the precision it was packed for; a call at another precision refuses it on the host
    from dcvgan_amd.native import WPack
    assert lib.dcv_conv_effective_precision(ctypes.byref(g)) == 1 and lib.dcv_conv_effective_precision(ctypes.byref(g_bf)) == 2
    fake = ctypes.create_string_buffer(64)      # never dereferenced: the stamp check comes before any launch
    pk = WPack(ctypes.addressof(fake), 1 << 20, 1, 2)
    assert lib.dcv_conv_forward(ctypes.byref(g), ctypes.addressof(fake), ctypes.byref(x), ctypes.addressof(fake), ctypes.addressof(fake), ctypes.byref(y_ok), 0, 0.0,
                                ctypes.byref(pk), None, 0, None) == -1 and b"dcv_wpack.precision" in lib.dcv_last_error()
    pk0 = WPack(ctypes.addressof(fake), 1 << 20, 0, 0)
    assert lib.dcv_conv_forward(ctypes.byref(g), ctypes.addressof(fake), ctypes.byref(x), ctypes.addressof(fake), ctypes.addressof(fake), ctypes.byref(y_ok), 0, 0.0,
                                ctypes.byref(pk0), None, 0, None) == -1 and b"dcv_wpack.precision" in lib.dcv_last_error()


def test_product_path_has_no_cpu_fallback(lib):
    import torch
    from dcvgan_amd import native, ops
    x = torch.zeros(1, 3, 8, 8)
    w = torch.zeros(4, 3, 4, 4)
    with pytest.raises(native.NativeError):
        ops.conv(x, w, ops.conv_geom(w, (2, 2), (1, 1), False))


def test_no_product_import_of_oracle():
    """Nothing under dcvgan_amd/ may import the oracle (it is test infrastructure)."""
    for dp, _, files in os.walk(os.path.join(ROOT, "dcvgan_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), os.path.join(dp, f)


def test_library_has_no_packed_fp32_arithmetic(tmp_path):
    """dcvgan_amd/csrc/build.sh passes -target-feature -packed-fp32-ops (profiles/r04_packed_fp32/SUMMARY.txt: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 code in
    the small kernels gave wrong upper lanes beside another stream's bf16-MFMA waves).  Checked on the shipped code objects themselves."""
    import re
    import shutil
    import subprocess
    from dcvgan_amd import native
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    lib = str(tmp_path / "lib.so")
    shutil.copy(native.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", lib], capture_output=True, text=True, cwd=str(tmp_path))     # writes lib.so.<n>.hipv4-amdgcn-amd-amdhsa--gfx950 beside it
    objs = sorted(str(f) for f in tmp_path.iterdir() if "amdgcn" in f.name and "gfx950" in f.name)
    assert objs, "no gfx950 code object found in the library"
    mfma = 0
    for o in objs:
        asm = subprocess.run([objdump, "-d", "--mcpu=gfx950", o], capture_output=True, text=True).stdout
        mfma += len(re.findall(r"\bv_mfma_", asm))
        found = sorted(set(re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", asm)))
        assert not found, f"packed-FP32 arithmetic in a shipped kernel ({os.path.basename(o)}): {found}"
    assert mfma > 1000, "the disassembly does not look like the library's kernels"
