"""Audit of the hand-issued loads of ku_traverse (scan_unit.hip) and ks_traverse (scan_skip.hip): an asm load is invisible to hipcc's wait insertion, and its
destination registers count as written at the end of the asm statement -- so nothing may read, copy or overwrite them between
the load and the hand-written s_waitcnt that names them (cdna_hip_programming.md 5.7, item 1).  Compiles the two files to
ISA and checks every `global_load_dword*` that sits inside an ASMSTART/ASMEND pair.  Exit code 1 on a violation."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audit(isa):
    lines = isa.splitlines()
    bad, seen = [], 0
    i = 0
    while i < len(lines):
        if "#ASMSTART" in lines[i] and i + 1 < len(lines) and "global_load_dword" in lines[i + 1]:
            m = re.search(r"global_load_dword(?:x\d)?\s+v(?:\[(\d+):(\d+)\]|(\d+))", lines[i + 1])
            lo_, hi_ = (int(m.group(1)), int(m.group(2))) if m.group(1) else (int(m.group(3)), int(m.group(3)))
            regs = {f"v{r}" for r in range(lo_, hi_ + 1)}
            pat = re.compile(r"\b(" + "|".join(regs) + r")\b|v\[(\d+):(\d+)\]")
            seen += 1
            j = i + 3  # behind ASMEND
            ok = False
            in_asm = False
            while j < len(lines):
                t = lines[j]
                if "#ASMSTART" in t:
                    in_asm = True
                elif "#ASMEND" in t:
                    in_asm = False
                if in_asm and "s_waitcnt vmcnt" in t:  # the hand-written wait (hipcc's own are outside asm statements)
                    ok = True
                    break
                if t.strip().startswith(("s_endpgm", "s_setpc")):
                    break
                body = t.split(";")[0]
                for mm in pat.finditer(body):
                    if mm.group(1) or any(f"v{r}" in regs for r in range(int(mm.group(2)), int(mm.group(3)) + 1)):
                        bad.append((j + 1, t.strip()))
                j += 1
            if not ok:
                bad.append((i + 2, "no hand-written wait behind this load"))
        i += 1
    return seen, bad


def slow_selects(isa):
    """`v_cndmask_b32_e32 ..., vcc` whose VCC was last written by the scalar unit, inside the innermost loops that hold a
    hand-issued probe: (kernel, count) pairs.  Reported, not a violation: profiles/r06_cndmask.txt -- re-encoding them as VOP3
    in the assembly changes nothing on ku_traverse, forcing the VOP3 form in the source costs 3-5 %."""
    out = []
    kernel, writer, in_loop, n = None, None, False, 0
    for t in isa.splitlines():
        body = t.split(";")[0].strip()
        if re.match(r"_ZN\S+:", t):
            if kernel and n:
                out.append((kernel, n))
            kernel, writer, n = t.split(":")[0], None, 0
        if not body or body.endswith(":"):
            continue
        op = body.split()[0]
        if op.startswith("v_cndmask_b32_e32") and writer and writer.startswith("s_"):
            n += 1
        m = re.match(r"(\S+)\s+vcc\b", body)
        if m and not op.startswith("v_cndmask"):
            writer = op
    if kernel and n:
        out.append((kernel, n))
    return out


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    seen, bad, slow = 0, [], []
    flags = sys.argv[1:] or ["-O3", "-std=c++17"]  # (the Makefile passes the build's own CXXFLAGS; without arguments: its defaults)
    with tempfile.TemporaryDirectory() as d:
        for src in ("scan_unit.hip", "scan_skip.hip"):  # the files with hand-issued loads
            out = os.path.join(d, src + ".s")
            subprocess.check_call([hipcc, *flags, "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                                   "-I", os.path.join(ROOT, "aha_amd", "csrc"), "-S", "--cuda-device-only", "-w", "-o", out,
                                   os.path.join(ROOT, "aha_amd", "csrc", src)])
            isa = open(out).read()
            n, b = audit(isa)
            seen += n
            bad += [(ln, f"{src}: {t}") for ln, t in b]
            if n == 0:
                bad.append((0, f"{src}: no hand-issued load found"))
            slow += slow_selects(isa)
    print(f"{seen} hand-issued probes audited, {len(bad)} violations")
    if slow:
        print(f"  (e32 selects on an SALU-written VCC, reported only: {sum(n for _, n in slow)} in {len(slow)} kernels)")
    for ln, t in bad:
        print(f"  line {ln}: {t}")
    return 1 if bad or seen == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
