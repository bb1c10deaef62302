"""The C-ABI library loads on a machine without a GPU, exports every function
include/alore_nmpc.h declares, and refuses to create a solver without a device
(no CPU fallback behind the ABI)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "alore_nmpc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(alore_nmpc_[a-z_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("alore_nmpc_create", "alore_nmpc_destroy", "alore_nmpc_rti", "alore_nmpc_linearize",
                 "alore_nmpc_forward_simulate", "alore_nmpc_shift", "alore_nmpc_batch_alloc",
                 "alore_nmpc_batch_upload", "alore_nmpc_batch_download", "alore_nmpc_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from alore_legged_manipulator_amd import _lib
    lib = _lib.load()
    bound = {n for n, _, _ in _lib.SYMBOLS}
    for name in declared_functions():
        assert hasattr(lib, name), f"libalore_nmpc.so does not export {name}"
        assert name in bound, f"{name} is declared in the header but not bound in _lib.SYMBOLS"


def test_backend_and_host_headers_are_exported_too():
    """include/alore_backend.h (libalore_nmpc.so) and include/alore_nmpc_host.h (libalore_nmpc_host.so)"""
    from alore_legged_manipulator_amd import _lib, backend
    lib = _lib.load()
    backend._bind(lib)
    for hdr, prefix, so in (("alore_backend.h", "alore_backend_", lib),):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", hdr)).read(), flags=re.S)
        names = sorted(set(re.findall(r"\b(%s[a-z_]+)\s*\(" % prefix, src)))
        assert len(names) >= 12
        for n in names:
            assert hasattr(so, n), f"{n} declared in {hdr} but not exported"
    host_so = os.path.join(ROOT, "alore_legged_manipulator_amd", "libalore_nmpc_host.so")
    syms = os.popen(f"nm -D --defined-only {host_so}").read()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "alore_nmpc_host.h")).read(), flags=re.S)
    for n in sorted(set(re.findall(r"\b(alore_host_[a-z_]+)\s*\(", src))):
        assert f" {n}\n" in syms, f"{n} declared in alore_nmpc_host.h but not exported"
    # ctypes mirrors have the C sizes (checked against a C compiler)
    import subprocess, tempfile
    code = '#include <stdio.h>\n#include "alore_backend.h"\nint main(){printf("%zu %zu %zu", sizeof(alore_backend_config), sizeof(alore_backend_status), sizeof(alore_flat_traj));}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(code)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert sizes == [C.sizeof(backend.BackendConfig), C.sizeof(backend.StatusC), C.sizeof(backend.FlatTrajC)]


def test_no_backend_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from alore_legged_manipulator_amd.backend import BackendError, BatchedMSPlanner
    with pytest.raises(BackendError):
        BatchedMSPlanner(4, 16)


def test_no_cpu_fallback_behind_the_abi():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from alore_legged_manipulator_amd import _lib
    lib = _lib.load()
    cfg = _lib.Config(20, 0.01, 0, 0, 0, -1)
    h = C.c_void_p()
    rc = lib.alore_nmpc_create(C.byref(cfg), C.byref(h))
    assert rc == -2 and not h.value  # ALORE_NMPC_E_NO_DEVICE
    from alore_legged_manipulator_amd.nmpc import BatchedNmpc
    with pytest.raises(Exception):
        BatchedNmpc(4, 20)


def test_struct_layouts_match_the_header():
    """ctypes mirrors of the ABI structs: member order must follow the header."""
    from alore_legged_manipulator_amd import _lib
    src = open(os.path.join(ROOT, "include", "alore_nmpc.h")).read()
    body = re.search(r"typedef struct \{(.*?)\} alore_nmpc_batch;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    members = re.findall(r"\*\s*([A-Za-z0-9_]+)\s*;", body)
    assert tuple(members) == _lib.BATCH_MEMBERS
    cfg = re.search(r"typedef struct \{(.*?)\} alore_nmpc_config;", src, flags=re.S).group(1)
    cfg = re.sub(r"/\*.*?\*/", "", cfg, flags=re.S)
    cfg_members = re.findall(r"(?:int|float)\s+([A-Za-z0-9_]+)\s*;", cfg)
    assert cfg_members == [n for n, _ in _lib.Config._fields_]


def test_compat_library_exports_the_model_functions():
    """acado_rhs / acado_diffs of libalore_acado_compat.so (host, no GPU needed) against the oracle's restatement
    and, where it was built, the compiled reference: bit-exact."""
    import numpy as np
    pkg = os.path.join(ROOT, "alore_legged_manipulator_amd")
    import subprocess, tempfile
    C.CDLL(os.path.join(pkg, "libalore_nmpc.so"), mode=C.RTLD_GLOBAL)
    # the two process-global structs are the CALLER's (mpc_wrapper.cpp:27-28 defines them in the reference)
    tmp = tempfile.mkdtemp()
    open(os.path.join(tmp, "g.c"), "w").write('#include "alore_acado_compat.h"\nACADOvariables acadoVariables;\nACADOworkspace acadoWorkspace;\n')
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), os.path.join(tmp, "g.c"), "-o", os.path.join(tmp, "libg.so")])
    C.CDLL(os.path.join(tmp, "libg.so"), mode=C.RTLD_GLOBAL)
    S = C.CDLL(os.path.join(pkg, "libalore_acado_compat.so"))
    O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_nmpc.so"))
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libacado_ref.so")
    R = C.CDLL(ref_so) if os.path.exists(ref_so) else None
    FP = C.POINTER(C.c_float)
    rng = np.random.default_rng(12)
    for _ in range(200):
        xin = np.concatenate([rng.uniform(-5, 5, 2), rng.uniform(-7, 7, 1), rng.uniform(-20, 20, 2),
                              [rng.uniform(-0.3, 0.3), rng.uniform(-0.5, -0.1), rng.uniform(0.1, 0.5)]]).astype(np.float32)
        outs = []
        for lib, names in ((S, ("acado_rhs", "acado_diffs")), (O, ("orc_rhs", "orc_diffs")), (R, ("acado_rhs", "acado_diffs"))):
            if lib is None:
                continue
            r, d = np.zeros(3, np.float32), np.zeros(15, np.float32)
            getattr(lib, names[0])(xin.ctypes.data_as(FP), r.ctypes.data_as(FP))
            getattr(lib, names[1])(xin.ctypes.data_as(FP), d.ctypes.data_as(FP))
            outs.append(np.concatenate([r, d]))
        for o in outs[1:]:
            assert np.array_equal(outs[0], o)


@pytest.mark.parametrize("hdr,prefix,minimum", [("alore_ltv_mpc.h", "alore_ltv_", 8), ("alore_wb.h", "alore_wb_", 14)])
def test_ltv_and_whole_body_headers_are_exported(hdr, prefix, minimum):
    from alore_legged_manipulator_amd import _lib
    lib = _lib.load()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", hdr)).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(%s[a-z_]+)\s*\(" % prefix, src)))
    assert len(names) >= minimum
    for n in names:
        assert hasattr(lib, n), f"{n} declared in {hdr} but not exported"


def test_no_whole_body_or_ltv_solver_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from alore_legged_manipulator_amd.ltv_mpc import BatchedLtvMpc, LtvError
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody, WbError
    with pytest.raises(WbError):
        BatchedWholeBody(4)
    with pytest.raises(LtvError):
        BatchedLtvMpc(4)


def test_whole_body_kernels_stay_below_their_residency_cliffs():
    """LDS per workgroup decides how many workgroups a CU holds: one more 256-byte block in the Riccati kernel (or a
    __syncthreads_or, which owns an LDS temporary) costs a third of its throughput (measured: 2.8 -> 4.2 ms)."""
    from alore_legged_manipulator_amd import _lib
    from alore_legged_manipulator_amd.whole_body import _bind
    lib = _lib.load()
    _bind(lib)
    a, b = C.c_int(), C.c_int()
    assert lib.alore_wb_kernel_info(C.byref(a), C.byref(b)) == 0
    assert a.value <= 20480, a.value       # 8 wavefronts of the stage kernel per CU
    assert b.value <= 53760, b.value       # 3 workgroups of the Riccati kernel per CU
