"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/stage_hip.h
declares, the ctypes signatures mirror the header, and the STAGE class has the reference's parameter schema."""
import os
import re
import subprocess

import pytest
import torch

from conftest import MODEL_CASES, ROOT, Fixture

HEADER = os.path.join(ROOT, "include", "stage_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t|const char\*|void\*|void|float)\s+(stage_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).split(",")]
        if args == ["void"]:
            args = []
        out[m.group(2)] = (m.group(1), args)
    return out


@pytest.fixture(scope="module")
def built_lib():
    from tvqaplus_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", ROOT, "-j8"], stdout=subprocess.DEVNULL)
    return _lib


def test_header_symbols_exported(built_lib):
    lib = built_lib.load()
    decl = _declared()
    assert len(decl) >= 24
    for name in decl:
        assert hasattr(lib, name), name
    # one number in three places: the header's macro (what the library returns), the Python binding's constant, this test
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "stage_hip.h")).read()
    assert int(re.search(r"#define STAGE_HIP_ABI_VERSION (\d+)", hdr).group(1)) == built_lib.ABI_VERSION == lib.stage_hip_abi_version() == 5
    assert lib.stage_hip_error_string(-1).decode().startswith("stage_hip")


def test_ctypes_signatures_match_header(built_lib):
    import ctypes
    decl = _declared()
    assert set(decl) == set(built_lib.SIGNATURES), set(decl) ^ set(built_lib.SIGNATURES)

    def kind(c_arg: str):
        c_arg = c_arg.strip()
        if "*" in c_arg:
            return ctypes.c_void_p
        base = re.sub(r"\b\w+$", "", c_arg).strip() or c_arg  # drop the parameter name
        return {"int": ctypes.c_int, "float": ctypes.c_float, "long long": ctypes.c_longlong,
                "unsigned long long": ctypes.c_ulonglong, "size_t": ctypes.c_size_t}[base]

    for name, (ret, args) in decl.items():
        res, argtypes = built_lib.SIGNATURES[name]
        assert len(args) == len(argtypes), name
        for i, (a, t) in enumerate(zip(args, argtypes)):
            assert kind(a) is t, (name, i, a, t)
        exp_ret = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p, "void*": ctypes.c_void_p,
                   "void": None, "float": ctypes.c_float}[ret]
        assert res is exp_ret, name


def test_workspace_queries_need_no_gpu(built_lib):
    lib = built_lib.load()
    assert lib.stage_ln_bwd_ws_bytes(384) >= 2 * 384 * 4
    assert lib.stage_gemm_tn_ws_bytes(960000, 128, 384) >= 128 * 384 * 4
    assert lib.stage_dwconv_bwd_ws_bytes(128, 7) >= 8 * 128 * 4
    assert lib.stage_str_attn_bwd_ws_bytes(16, 5, 40, 128) >= 16 * 5 * 40 * 128 * 4


@pytest.mark.parametrize("name", MODEL_CASES)
def test_state_dict_schema_matches_reference(name):
    """Keys, shapes and buffer values of the drop-in class == the reference's state_dict stored in the fixture."""
    from tvqaplus_amd.stage import STAGE
    fx = Fixture(name)
    model = STAGE(fx.opt)
    ref = fx.group("param")
    sd = model.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        if k.endswith("position_encoding.pe"):
            assert torch.equal(sd[k], v), k
    model.load_state_dict(ref, strict=True)
    assert model.inference_mode is False and model.num_a == 5 and model.bridge_hsz == 300


def test_product_refuses_cpu_tensors():
    from tvqaplus_amd import ops
    from tvqaplus_amd._lib import StageHipError
    with pytest.raises(StageHipError):
        ops.layernorm(torch.randn(4, 16), torch.ones(16), torch.zeros(16))
    with pytest.raises(StageHipError):
        ops.structured_attention(torch.randn(1, 5, 4, 16), torch.randn(1, 2, 3, 16), torch.ones(1, 5, 4),
                                 torch.ones(1, 2, 3), 10.0)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under tvqaplus_amd/ may reference it."""
    pkg = os.path.join(ROOT, "tvqaplus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "stage_oracle" not in src, f


def test_kernels_with_untracked_loads_do_not_spill(built_lib):
    """Kernels that issue inline-asm loads and await them with hand-counted s_waitcnt must not spill: a spilled
    register whose load is still in flight stores stale data (finding 3).  The streaming GEMMs, the K1 forward
    kernels that run at the published shapes and their callers' objects are checked through the code-object metadata."""
    import glob
    import re
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("ROCm LLVM tools not available")
    # kernel-name pattern -> must have vgpr_spill_count == 0
    must_be_clean = {
        "gemm_stream": [r"gemm_nt_stream_kernel", r"gemm_tn_stream_kernel", r"gemm_tn_share_kernel", r"gemm_tn_wide_kernel"],
        # video-stream shapes of the published configs (RT = 2, KL = 1, PERM), eval and training variants
        "str_attn_fwd_reg": [r"str_attn_fwd_reg_kernelILi2ELi1ELb1ELb0E"],
    }
    for stem, patterns in must_be_clean.items():
        obj = os.path.join(ROOT, "build", stem + ".o")
        if not os.path.exists(obj):
            subprocess.check_call(["make", "-C", ROOT, "build/%s.o" % stem], stdout=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(llvm, "llvm-objdump"), "--offloading", obj], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        bundles = glob.glob(obj + ".*amdgcn*")
        assert bundles, "no device code object in " + obj
        try:
            notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", bundles[0]]).decode()
        finally:
            for f in glob.glob(obj + ".0.*"):
                os.remove(f)
        names = re.findall(r"\.name:\s+(\S+)", notes)
        spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", notes)
        assert len(names) == len(spills) and names
        for pat in patterns:
            hits = [(n, int(s)) for n, s in zip(names, spills) if re.search(pat, n)]
            assert hits, "no kernel matches " + pat
            for n, sp in hits:
                assert sp == 0, "%s spills %d VGPRs" % (n, sp)


def test_cpp_host_example_builds_against_the_c_abi(tmp_path):
    """examples/k1_forward_host.cpp: a C++ host with nothing but the HIP runtime and include/stage_hip.h compiles and links
    against the library (no GPU needed to link; tests/test_hip_ops.py runs it)."""
    import shutil, subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "k1_forward_host")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "k1_forward_host.cpp"), "-L", os.path.join(root, "tvqaplus_amd"),
                           "-lstage_hip", "-o", exe])
    assert os.path.getsize(exe) > 0


def test_model_stage_import_shim():
    """The reference drivers' own import line (main.py:13, inference.py:7: ``from model.stage import STAGE``) resolves to the
    HIP class when ``<repo>/shim`` leads PYTHONPATH -- in a fresh interpreter, as a driver process would see it."""
    import subprocess
    import sys
    code = ("from model.stage import STAGE; import tvqaplus_amd.stage as S; assert STAGE is S.STAGE; "
            "from tvqaplus_amd.synth import make_opt; import contextlib, io\n"
            "with contextlib.redirect_stdout(io.StringIO()): m = STAGE(make_opt(hsz=16, embedding_size=8, vfeat_size=8))\n"
            "assert m.num_a == 5 and m.bridge_hsz == 300 and m.inference_mode is False; print('shim ok')")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "shim"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and "shim ok" in out.stdout, out.stderr
