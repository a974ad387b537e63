"""The C-ABI shared library: loads without a GPU, exports every symbol the header declares,
validates arguments, and refuses to compute without a device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import nabo_amd
from nabo_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(REPO, "include", "nabo_knn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nabo_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_header_symbols():
    L = _lib.lib()
    names = _header_functions()
    assert len(names) >= 15
    assert sorted(names) == sorted(_lib.SYMBOLS)
    for n in names:
        assert hasattr(L, n), "libnabo_knn.so does not export %s" % n
    assert b"gfx950" in L.nabo_version()


def test_argument_validation_maps_to_value_error():
    x = np.zeros((4, 3))
    with pytest.raises(ValueError):
        nabo_amd.knn(x, np.zeros((5, 2)), 2)            # component mismatch
    with pytest.raises(ValueError):
        nabo_amd.knn(x, np.zeros((5, 3)), 2, ref_mask=np.zeros(4, np.uint8))   # mask length


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    if nabo_amd.device_count() > 0:
        pytest.skip("a GPU is visible here; the no-device path is covered on the CPU box")
    x = np.random.default_rng(0).standard_normal((8, 4))
    with pytest.raises(nabo_amd.NaboError) as e:
        nabo_amd.knn(x, x, 2)
    assert "no HIP device" in str(e.value)
    with pytest.raises(nabo_amd.NaboError):
        nabo_amd.pairwise(x, x)


def test_product_never_imports_the_oracle():
    """Nothing under nabo_amd/ may import, load or link oracle/ (it is test infrastructure)."""
    pkg = os.path.join(REPO, "nabo_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libnabo_oracle" not in txt, f


def _build_c_consumer(tmp_path):
    import subprocess
    exe = os.path.join(str(tmp_path), "abi_check")
    cmd = ["gcc", "-std=c99", "-Wall", "-I" + os.path.join(REPO, "include"), os.path.join(REPO, "tests", "abi_c", "abi_check.c"),
           "-L" + os.path.join(REPO, "nabo_amd"), "-lnabo_knn", "-Wl,-rpath," + os.path.join(REPO, "nabo_amd"),
           "-Wl,-rpath-link,/opt/rocm/lib", "-lm", "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_header_is_plain_c_and_every_entry_point_links(tmp_path):
    """include/nabo_knn.h through gcc -std=c99 (the boundary a cgo / JNI / ctypes binding sees), linked against
    the library; the program only takes the addresses and prints the version."""
    import subprocess
    _lib.lib()
    exe = _build_c_consumer(tmp_path)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert r.returncode == 0 and "%d entry points" % len(_lib.SYMBOLS) in r.stdout, r.stdout


@pytest.mark.gpu
def test_plain_c_consumer_gets_reference_results(tmp_path):
    import subprocess
    _lib.lib()
    exe = _build_c_consumer(tmp_path)
    r = subprocess.run([exe, "run"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert r.returncode == 0 and "C ABI ok" in r.stdout, r.stdout


def test_inline_assembly_mfma_kernels_keep_accumulators_out_of_agprs():
    """l2c_topk.hip issues its MFMAs from inline assembly (hand-scheduled with the filter's instructions): hipcc neither knows
    their latency nor pads their hazards, so it must never touch an accumulator right behind them -- which it does as soon as
    it parks accumulators in AGPRs (v_accvgpr_write / _read copies; seen at two waves per SIMD with the B operands pinned in
    AGPRs: wrong neighbours).  Compile the file to assembly and require that the hand-scheduled kernels (two and four steps
    of 32 slots, every geometry) spill nothing and contain no AGPR copy in any basic block that issues an MFMA nor within 16
    instructions behind one -- AGPR parking around the drain CALL of the rare staging path (live registers moved out of the
    callee's way, at one wave per SIMD where all 256 arch VGPRs are live) is fine: at least 16 instructions and an
    explicit s_nop 15 lie between the tile's last MFMA and that path."""
    import re
    import subprocess
    from nabo_amd import _build
    src = os.path.join(_build.CSRC, "l2c_topk.hip")
    cmd = [_build.hipcc()] + _build.FLAGS + _build.FILE_FLAGS["l2c_topk.hip"] + ["-S", "--cuda-device-only", src, "-o", "-"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = r.stdout
    seen = 0
    for m in re.finditer(r"^(_ZN4nabo15l2c_topk_kernelILi([24])E\w+):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
        name, body = m.group(1), m.group(3)
        seen += 1
        assert body.count("v_mfma_f32_16x16x32_f16") >= 48, name
        since_mfma, block_has_mfma, block_has_acc = 10 ** 9, False, False
        for line in body.split("\n"):
            t = line.strip()
            if not t or t.startswith(";"):
                continue
            if re.match(r"^\.?\w+:", t):                      # a label: a new basic block
                assert not (block_has_mfma and block_has_acc), name
                block_has_mfma = block_has_acc = False
                continue
            if t.startswith("v_mfma"):
                since_mfma, block_has_mfma = 0, True
                continue
            since_mfma += 16 if t.startswith("s_nop 15") else 1
            if t.startswith("v_accvgpr"):
                block_has_acc = True
                assert since_mfma > 16, (name, t)
        assert not (block_has_mfma and block_has_acc), name
        assert "scratch_" not in body, name
    assert seen == 5, seen        # two steps: geometries A, B, C; four steps: A, C
    for m in re.finditer(r"\.name:\s+(_ZN4nabo15l2c_topk_kernel\w+)\n(?:.*\n){1,12}?\s+\.vgpr_spill_count:\s+(\d+)", asm):
        assert int(m.group(2)) == 0, m.group(0)
    # ... and, register by register (round-3 advisory): hipcc brackets inline assembly with ;;#ASMSTART / ;;#ASMEND, so the
    # destination quads of the hand-issued MFMAs are known exactly.  No instruction hipcc itself placed -- a v_mov, a spill
    # copy, a load, anything outside an assembly statement -- may read or write one of them within 16 instruction slots of
    # the MFMA that writes it (12 wait states for this shape; s_nop N counts N + 1).  A toolchain update that starts to
    # touch accumulators right behind the statements fails HERE, on the CPU box, not as wrong neighbours on the GPU.
    def regs(tok):
        out = set()
        for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
            out.update(range(int(a), int(b) + 1))
        out.update(int(a) for a in re.findall(r"\bv(\d+)\b", tok))
        return out
    checked = 0
    for m in re.finditer(r"^(_ZN4nabo15l2c_topk_kernelILi([24])E\w+):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
        name, body = m.group(1), m.group(3)
        in_asm, live = False, []                              # live: [registers of an asm MFMA's destination, its age]
        for line in body.split("\n"):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";") or re.match(r"^\.?\w+:", t):
                continue
            age = int(t.split()[1]) + 1 if t.startswith("s_nop") else 1
            if not in_asm:
                used = regs(t.split(";")[0])
                for dst, a in live:
                    assert not (used & dst), (name, t, sorted(dst), a)
                checked += 1
            live = [[d, a + age] for d, a in live if a + age <= 16]
            if in_asm and t.startswith("v_mfma"):
                live.append([regs(t.split(",")[0]), 0])
    assert checked > 1000
