"""C-ABI surface checks that need no GPU: the library builds/loads and exports exactly the
entry points include/pgca_hip.h declares; the ctypes struct mirrors the C struct."""
import ctypes
import os
import re
import subprocess

import pytest
import torch  # noqa: F401  (loads libamdhip64 first, as the product does)

from pgca_amd import REPO_ROOT, hip


def header_functions():
    src = open(os.path.join(REPO_ROOT, "include", "pgca_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pgca_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(hip.LIB_PATH):
        from pgca_amd import build
        build.build()
    return hip.load()


def test_every_declared_symbol_is_exported(lib):
    declared = header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in pgca_hip.h but not exported"
    assert sorted(hip.EXPORTS) == declared, set(hip.EXPORTS) ^ set(declared)


def test_version_and_error_string(lib):
    assert lib.pgca_version() >= 100
    assert isinstance(lib.pgca_last_error(), bytes)


def test_gemm_args_layout_matches_c_struct(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    fields = [f[0] for f in hip.GemmArgs._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "pgca_hip.h"\nint main(){printf("%zu", sizeof(pgca_gemm_args));'
    for f in fields:
        prog += f'printf(" %zu", offsetof(pgca_gemm_args, {f}));'
    prog += "return 0;}\n"
    c = tmp_path / "o.c"
    c.write_text(prog)
    exe = tmp_path / "o"
    subprocess.run(["gcc", "-I", os.path.join(REPO_ROOT, "include"), str(c), "-o", str(exe)], check=True)
    nums = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert nums[0] == ctypes.sizeof(hip.GemmArgs)
    assert nums[1:] == [getattr(hip.GemmArgs, f).offset for f in fields]


def test_skinny_args_layout_matches_c_struct(tmp_path):
    fields = [f[0] for f in hip.SkinnyArgs._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "pgca_hip.h"\nint main(){printf("%zu", sizeof(pgca_skinny_args));'
    for f in fields:
        prog += f'printf(" %zu", offsetof(pgca_skinny_args, {f}));'
    prog += "return 0;}\n"
    c = tmp_path / "s.c"
    c.write_text(prog)
    exe = tmp_path / "s"
    subprocess.run(["gcc", "-I", os.path.join(REPO_ROOT, "include"), str(c), "-o", str(exe)], check=True)
    nums = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert nums[0] == ctypes.sizeof(hip.SkinnyArgs)
    assert nums[1:] == [getattr(hip.SkinnyArgs, f).offset for f in fields]


def test_host_side_argument_validation_needs_no_gpu(lib):
    """Validation happens before any launch, so it is observable on a CPU-only box."""
    a = hip.GemmArgs()
    assert lib.pgca_gemm_bf16(ctypes.byref(a), None) == -1
    assert b"null operand" in lib.pgca_last_error()
    assert lib.pgca_attention_fwd(None, None, 1, 128, 1, 1, None, None, 0, 0, 1.0, None, None) == -1
    sk = hip.SkinnyArgs()
    assert lib.pgca_gemm_skinny(ctypes.byref(sk), None) == -1 and b"pgca_gemm_skinny" in lib.pgca_last_error()
    assert lib.pgca_gemm_skinny_workspace(1, 3072, 1024) > 0 and lib.pgca_gemm_skinny_workspace(0, 8, 8) == 0
    assert lib.pgca_seq_pack_prepare(None, 1, 16, 64, None, None, None, None, None, None, None) == -1
    assert lib.pgca_sqnorm_blocks(1) == 1 and lib.pgca_sqnorm_blocks(16384 * 3 + 1) == 4
