"""ISA lint of the hand-scheduled symmetric sweep (fortran_davidson_amd/csrc/k_matvec_symw.hip), no GPU needed.

The kernel's MFMAs are inline assembly (the compiler does not know them as such, so it neither pads their hazards nor keeps
out of the accumulation registers the kernel names literally).  The kernel header states the rules that make that safe; this
test checks the EMITTED code for them:
  1. no compiler-generated instruction (outside ;;#ASMSTART / ;;#ASMEND) names the kernel's fixed registers (a[128:255]: Gram-layout operands,
     load ring, X_J operand; the fp32-tile variant: v[224:255] too, its raw load ring);
  2. no VALU instruction writes a register that an MFMA reads within the next two instructions (2 wait states);
  3. a register written by an MFMA is read by a non-MFMA instruction only after an `s_nop 15` (+ `s_nop 3`): 18+ wait states -
     or after three later MFMAs of the wave (each holds the matrix pipe for 16 passes: the result is two MFMAs old at least);
  4. no scratch in the kernel at all, and no register moves between the halves (v_accvgpr_*) - except in the harness variants (GEN = 2 / 3,
     round 6), which run at the limit of the ordinary registers: there the compiler parks lane constants in accumulation registers, so
     these variants keep their fixed registers ABOVE a168 (symw_fixed_lo(true)) and rule 1 checks that the compiler stays below; their own
     v_accvgpr_read / _write of the table values stand inside asm statements;
  5. no buffer load reads a scalar register that a VALU instruction wrote fewer than 5 wait states before.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fortran_davidson_amd", "csrc", "k_matvec_symw.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
def fixed_lo(name):
    """first fixed accumulation register of a kernel variant (symw_fixed_lo in the kernel file): 128; the harness variants 168"""
    return 168 if _is_harness(name) else 128


def _is_harness(name):
    return re.search(r"ELi([23])EEv", name) is not None


def _regs(tok):
    """('v' | 'a', set of register numbers) named by an operand token such as v[10:13], a7, v3"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None, set()


def _operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else "", []
    return parts[0], [t.strip() for t in parts[1].replace(" offen", "").replace(" offset:", ",offset:").split(",")]


def _kernels(asm):
    out, cur, name = {}, None, None
    for ln in asm.splitlines():
        m = re.match(r"^(_Z\d+matvec_symw_kernel\w+):", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(ln)
            if "s_endpgm" in ln:
                out[name] = cur
                cur = None
    return out


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("isa") / "symw.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    "-o", str(out), SRC], check=True, capture_output=True, timeout=300)
    # -S does not run the assembler over the inline asm: assemble as well (an "s" operand the compiler could not keep in scalar
    # registers, a register name out of range, ... only show up there)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-c",
                    "-o", str(out) + ".o", SRC], check=True, capture_output=True, timeout=600)
    ks = _kernels(out.read_text())
    assert len(ks) == 7, list(ks)          # <1>, <2>, <1, TALL>, <1, F32>, <2, GEN = 1 (hashed), 2 (harness cos), 3 (harness sin)>
    assert sum(_is_harness(k) for k in ks) == 2
    return ks


def test_resources(kernels):
    for name, lines in kernels.items():
        text = "\n".join(lines)
        assert "scratch_" not in text, name
        if not _is_harness(name):
            assert "v_accvgpr" not in text, name


def test_compiler_code_stays_out_of_the_fixed_registers(kernels):
    for name, lines in kernels.items():
        lo = fixed_lo(name)
        in_asm = False
        for ln in lines:
            if "#ASMSTART" in ln:
                in_asm = True
                continue
            if "#ASMEND" in ln:
                in_asm = False
                continue
            if in_asm or ln.strip().startswith((";", ".")):
                continue
            _, ops = _operands(ln)
            for t in ops:
                f, r = _regs(t)
                assert not (f == "a" and r and max(r) >= lo), (name, ln)
                # the fp32-tile variant keeps its raw load ring in fixed VGPRs v[224:255] (SYMW_RAW in the kernel file)
                assert not ("ELb0ELb1ELb0E" in name and f == "v" and r and max(r) >= 224), (name, ln)


def test_mfma_hazards(kernels):
    for name, lines in kernels.items():
        code = [ln for ln in lines if ln.strip() and not ln.strip().startswith((";", ".")) and not ln.rstrip().endswith(":")]
        recent = []                      # (is_valu, file, regs written) of the last instructions
        mfma_written = {"v": {}, "a": {}}   # register -> index of the MFMA that wrote it, until drained
        n_mfma = 0
        for idx, ln in enumerate(code):
            op, ops = _operands(ln)
            if not op:
                continue
            if op.startswith("v_mfma"):
                n_mfma += 1
                dfile, dregs = _regs(ops[0])
                reads = [_regs(t) for t in ops[1:]]
                for wfile_w, wregs in [(f, r) for (isv, f, r) in recent[-2:] if isv]:
                    for rf, rr in reads:
                        assert not (rf == wfile_w and rr & wregs), (name, "VALU write -> MFMA read", ln)
                # a chain (D = C) is forwarded by the hardware; anything else reading a pending result is checked below
                for r in dregs:
                    mfma_written[dfile][r] = n_mfma
                recent.append((False, dfile, dregs))
                continue
            if op == "s_nop":
                if ops and ops[0] == "15":
                    mfma_written = {"v": {}, "a": {}}
                recent.append((False, None, set()))
                continue
            # non-MFMA instruction: must not read a register with an undrained MFMA result
            is_store = op.startswith(("ds_write", "global_store", "buffer_store"))
            srcs = ops if is_store else ops[1:]
            for t in srcs:
                f, r = _regs(t)
                if f:
                    pend = {x for x in r & set(mfma_written[f]) if n_mfma - mfma_written[f][x] < 3}
                    assert not pend, (name, "MFMA result read without drain", ln)
            is_valu = op.startswith("v_") and not op.startswith("v_mfma")
            dfile, dregs = _regs(ops[0]) if ops and not is_store else (None, set())
            # a non-MFMA write of a register (load, VALU) ends its "pending MFMA result" state
            if dfile:
                for r in dregs:
                    mfma_written[dfile].pop(r, None)
            recent.append((is_valu, dfile, dregs))
        assert n_mfma in (256, 512), (name, n_mfma)


def test_buffer_descriptors_are_not_fresh_from_the_valu(kernels):
    """A buffer instruction that reads a scalar register written by a VALU instruction (v_readfirstlane, v_readlane, v_cmp)
    needs 5 wait states in between.  The loads of the prologue open with `s_nop 4`; those of the loop do not (their descriptors
    are scalar-ALU results) - checked here: no VALU instruction among the five in front of a buffer load writes a scalar register
    that the load reads, unless an `s_nop 4` stands in between."""
    def sregs(tok):
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"s(\d+)", tok)
        return {int(m.group(1))} if m else ({-1} if tok == "vcc" else set())
    for name, lines in kernels.items():
        code = [ln.split(";")[0].strip() for ln in lines]
        code = [c for c in code if c and not c.startswith(".") and not c.endswith(":")]
        for i, c in enumerate(code):
            if not c.startswith("buffer_load"):
                continue
            _, ops = _operands(c)
            read = set().union(*[sregs(t.split()[0]) for t in ops[1:] if t]) if len(ops) > 1 else set()
            for b in reversed(code[max(0, i - 5):i]):
                if b == "s_nop 4":
                    break
                if b.startswith(("v_readfirstlane", "v_readlane", "v_cmp")):
                    _, bops = _operands(b)
                    assert not (sregs(bops[0]) & read), (name, b, c)


def test_small_kernels_keep_their_accumulators_in_place_and_their_loops_pipelined():
    """K2 / K3 (plain-HIP MFMA kernels, k_gram.hip / k_panel.hip) as the Makefile builds them: (1) the panel kernel moves nothing
    between the register halves (the compiler's default form copied all 64 accumulators around every 8 MFMAs: round 3); (2) in
    the pinned variant of its k loop the first wait of a round lets more loads stay in flight than one step issues - the loads
    of later steps are NOT waited for, i.e. the loop is a software pipeline; (3) its small-matrix operand comes in as 8-byte
    loads at a common stride (the operand image) - one per column tile and step; (4) no scratch anywhere."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    mk = open(os.path.join(ROOT, "fortran_davidson_amd", "csrc", "Makefile")).read()
    flags = re.search(r"FLAGS_k_panel\s*=\s*(.*)", mk).group(1).split()
    assert "-amdgpu-mfma-vgpr-form=1" in flags and re.search(r"FLAGS_k_gram\s*=\s*(.*)", mk).group(1).split() == flags
    for src, kernel_prefix in (("k_panel.hip", "_Z17panel_gemm_kernel"), ("k_gram.hip", "_Z11gram_kernel")):
        res = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), *flags, "-S",
                              "--cuda-device-only", os.path.join(ROOT, "fortran_davidson_amd", "csrc", src), "-o", "-"],
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        asm = res.stdout
        kernels = {}
        cur = None
        for ln in asm.splitlines():
            m = re.match(r"^(" + kernel_prefix + r"\w+):", ln)
            if m:
                cur = kernels.setdefault(m.group(1), [])
                continue
            if cur is not None:
                cur.append(ln)
                if "s_endpgm" in ln:
                    cur = None
        assert len(kernels) >= 3, list(kernels)
        for name, body in kernels.items():
            text = "\n".join(body)
            assert "scratch_" not in text, name
            if src == "k_panel.hip":
                assert "v_accvgpr" not in text, name
        assert ".private_segment_fixed_size: 0" in asm and not re.search(r"\.private_segment_fixed_size:\s+[1-9]", asm)
    # the pinned QT = 4 panel kernel: find its main loop (the inner loop with the most MFMAs)
    name = next(k for k in kernels_of_panel(ROOT, flags) if "ILi4ELi8ELb1" in k)
    body = kernels_of_panel(ROOT, flags)[name]
    loops, cur = [], None
    for ln in body:
        if "Inner Loop Header" in ln:
            cur = []
        if cur is not None:
            cur.append(ln)
            if re.search(r"s_cbranch_\w+\s+\.LBB", ln):
                loops.append(cur)
                cur = None
    main = max(loops, key=lambda l: sum("v_mfma_f64_16x16x4" in x for x in l))
    assert sum("v_mfma_f64_16x16x4" in x for x in main) == 64          # a round of 8 steps x 8 MFMAs
    assert sum("global_load_dwordx4" in x for x in main) == 8 and sum("global_load_dwordx2" in x for x in main) == 32
    waits = [int(m.group(1)) for x in main for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", x)] if m]
    assert waits and min(waits) >= 20, waits                           # never a drain: >= 4 steps of loads stay in flight


_PANEL_CACHE = {}


def kernels_of_panel(root, flags):
    if not _PANEL_CACHE:
        res = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"), *flags, "-S",
                              "--cuda-device-only", os.path.join(root, "fortran_davidson_amd", "csrc", "k_panel.hip"), "-o", "-"],
                             capture_output=True, text=True, timeout=600)
        cur = None
        for ln in res.stdout.splitlines():
            m = re.match(r"^(_Z17panel_gemm_kernel\w+):", ln)
            if m:
                cur = _PANEL_CACHE.setdefault(m.group(1), [])
                continue
            if cur is not None:
                cur.append(ln)
                if "s_endpgm" in ln:
                    cur = None
    return _PANEL_CACHE
