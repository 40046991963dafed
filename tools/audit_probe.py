"""Audit of the hand-issued probe of ku_traverse (scan_unit.hip): an asm load is invisible to hipcc's wait insertion, and its
destination registers count as written at the end of the asm statement -- so nothing may read, copy or overwrite them between
the load and the hand-written s_waitcnt that names them (cdna_hip_programming.md 5.7, item 1).  Compiles scan_unit.hip to
ISA and checks every `global_load_dwordx2` that sits inside an ASMSTART/ASMEND pair.  Exit code 1 on a violation."""
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


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "scan_unit.s")
        # (the Makefile passes the build's own CXXFLAGS; without arguments: its defaults)
        flags = sys.argv[1:] or ["-O3", "-std=c++17"]
        subprocess.check_call([hipcc, *flags, "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "aha_amd", "csrc"), "-S", "--cuda-device-only", "-w", "-o", out,
                               os.path.join(ROOT, "aha_amd", "csrc", "scan_unit.hip")])
        seen, bad = audit(open(out).read())
    print(f"{seen} hand-issued probes audited, {len(bad)} violations")
    for ln, t in bad:
        print(f"  line {ln}: {t}")
    return 1 if bad or seen == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
