"""Audit of the hand-issued loads of ku_traverse (scan_unit.hip), ks_traverse (scan_skip.hip) and k2d_expand_dense (scan_v2.hip): an asm load is invisible to hipcc's wait insertion, and its
destination registers count as written at the end of the asm statement -- so nothing may read, copy or overwrite them between
the load and the hand-written s_waitcnt that names them (cdna_hip_programming.md 5.7, item 1).  Compiles the two files to
ISA and checks every `global_load_dword*` that sits inside an ASMSTART/ASMEND pair.  Exit code 1 on a violation."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audit(isa):
    """Text order from every hand-issued load to the first hand-written wait behind it -- without the blocks that are laid
    out in between but cannot run there: a block behind an unconditional s_branch whose label no branch of the stretch
    itself names (hipcc places blocks of other paths of the loop wherever it likes)."""
    lines = isa.splitlines()
    bad, seen = [], 0
    br = re.compile(r"(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)")
    for i in range(len(lines) - 1):
        if not ("#ASMSTART" in lines[i] and "global_load_dword" in lines[i + 1]):
            continue
        m = re.search(r"global_load_dword(?:x\d)?\s+v(?:\[(\d+):(\d+)\]|(\d+))", lines[i + 1])
        lo_, hi_ = (int(m.group(1)), int(m.group(2))) if m.group(1) else (int(m.group(3)), int(m.group(3)))
        regs = {f"v{r}" for r in range(lo_, hi_ + 1)}
        pat = re.compile(r"\b(" + "|".join(regs) + r")\b|v\[(\d+):(\d+)\]")
        seen += 1
        if "s_waitcnt vmcnt(0)" in lines[i + 2]:  # the load and its wait in ONE asm statement
            continue
        # the stretch: up to the first hand-written wait
        end, in_asm = None, False
        for j in range(i + 3, len(lines)):
            t = lines[j]
            if "#ASMSTART" in t:
                in_asm = True
            elif "#ASMEND" in t:
                in_asm = False
            if in_asm and "s_waitcnt vmcnt" in t:
                end = j
                break
            if t.strip().startswith(("s_endpgm", "s_setpc")):
                break
        if end is None:
            bad.append((i + 2, "no hand-written wait behind this load"))
            continue
        targets = {mb.group(2) for t in lines[i + 3:end] for mb in [br.search(t.split(";")[0])] if mb}
        live, in_asm = True, False
        for j in range(i + 3, end):
            t = lines[j]
            body = t.split(";")[0].strip()
            if "#ASMSTART" in t:
                in_asm = True
            elif "#ASMEND" in t:
                in_asm = False
            ml = re.match(r"(\.LBB\d+_\d+):", t)
            if ml and not live:
                live = ml.group(1) in targets
            if not live or not body:
                continue
            if not (in_asm and "global_load_dword" in body):  # (the other loads of the group have registers of their own)
                for mm in pat.finditer(body):
                    if mm.group(1) or any(f"v{r}" in regs for r in range(int(mm.group(2)), int(mm.group(3)) + 1)):
                        bad.append((j + 1, f"{t.strip()}   <- register of the load at line {i + 2}, before its wait"))
            if body.startswith("s_branch"):
                live = False
    return seen, sorted(set(bad))


def sgpr_hazards(isa):
    """A VMEM instruction inside an asm statement that reads, as its scalar base, an SGPR a VALU instruction (v_readfirstlane,
    v_readlane) wrote fewer than five wait states before: hipcc pads this hazard for its own instructions only."""
    lines = [t.split(";")[0].strip() for t in isa.splitlines()]
    raw = isa.splitlines()
    bad = []
    in_asm = False
    for i, t in enumerate(lines):
        if "#ASMSTART" in raw[i]:
            in_asm = True
        elif "#ASMEND" in raw[i]:
            in_asm = False
        m = re.match(r"global_(?:load|store)_\w+\s+.*\bs\[(\d+):(\d+)\]", t)
        if not (in_asm and m):
            continue
        pair = {f"s{r}" for r in range(int(m.group(1)), int(m.group(2)) + 1)}
        states, j = 0, i - 1
        while j >= 0 and states < 5:
            u = lines[j]
            j += -1
            if not u or u.endswith(":"):
                continue
            mw = re.match(r"v_read(?:first)?lane_b32\s+(s\d+)", u)
            if mw and mw.group(1) in pair:
                bad.append((i + 1, f"{t}   <- {mw.group(1)} written by `{u}` {states} wait states before"))
                break
            mn = re.match(r"s_nop\s+(\d+)", u)
            states += int(mn.group(1)) + 1 if mn else 1
    return bad


def store_data_hazards(isa):
    """An asm store of more than 8 bytes whose data registers the very next instruction writes (one wait state on gfx9xx;
    hipcc pads it for its own stores only)."""
    raw = isa.splitlines()
    lines = [t.split(";")[0].strip() for t in raw]
    bad, in_asm = [], False
    for i, t in enumerate(lines):
        if "#ASMSTART" in raw[i]:
            in_asm = True
        elif "#ASMEND" in raw[i]:
            in_asm = False
        m = re.match(r"global_store_dwordx[34]\s+\S+\s+v\[(\d+):(\d+)\]", t)
        if not (in_asm and m):
            continue
        data = set(range(int(m.group(1)), int(m.group(2)) + 1))
        j = i + 1
        while j < len(lines) and (not lines[j] or lines[j].endswith(":")):
            j += 1
        if j >= len(lines) or lines[j].startswith("s_"):
            continue  # (a scalar instruction, s_nop among them, is the wait state)
        md = re.match(r"\S+\s+v(?:\[(\d+):(\d+)\]|(\d+))", lines[j])
        if md:
            dst = set(range(int(md.group(1)), int(md.group(2)) + 1)) if md.group(1) else {int(md.group(3))}
            if dst & data and not lines[j].startswith(("global_store", "ds_write", "buffer_store")):
                bad.append((j + 1, f"{lines[j]}   <- writes data registers of the asm store at line {i + 1} in the next instruction"))
    return bad


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
        for src in ("scan_unit.hip", "scan_skip.hip", "scan_v2.hip"):  # the files with hand-issued loads (scan_v2.hip: k2d_expand_dense)
            out = os.path.join(d, src + ".s")
            subprocess.check_call([hipcc, *flags, "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                                   "-I", os.path.join(ROOT, "aha_amd", "csrc"), "-S", "--cuda-device-only", "-w", "-o", out,
                                   os.path.join(ROOT, "aha_amd", "csrc", src)])
            isa = open(out).read()
            n, b = audit(isa)
            seen += n
            bad += [(ln, f"{src}: {t}") for ln, t in b + sgpr_hazards(isa) + store_data_hazards(isa)]
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
