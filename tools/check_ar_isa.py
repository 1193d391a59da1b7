#!/usr/bin/env python3
"""Static check of the hand-counted waits of k_gemm_ar (csrc/gemm_ar.hip) on the ISA hipcc emits for gfx950.

The kernel issues its A loads as inline asm, so the compiler inserts no s_waitcnt for them: the kernel's own counted
`s_waitcnt vmcnt(N)` in front of every stage barrier is all that stands between a load and the first use of its
registers.  This script compiles the file to assembly (no GPU needed), walks every k_gemm_ar kernel in program order --
the prologue once, then the loop body TWICE (the second pass starts with the first pass's requests still in flight)
-- with a model of the vector-memory counter (operations retire in issue order; `s_waitcnt vmcnt(N)` leaves the N
youngest outstanding) and reports
  * any instruction that reads or overwrites a register a global_load still owns,
  * any `s_waitcnt vmcnt(0)` inside the loop (a drained prefetch ring),
  * the vector-memory operations between consecutive barriers of the loop (expected: B_PW LDS-DMA pieces, then 4 loads).
Exit code 0 = clean.  usage: check_ar_isa.py [--keep /tmp/gemm_ar.s] [--define MACRO ...]
(`--define GS_AR_STAMPS` checks the probe build, which is NOT clean: the case this script was written for.)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gnn-epc-saft_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def compile_isa(path, defines=()):
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{ROOT}/include", "-S",
           "--cuda-device-only", os.path.join(CSRC, "gemm_ar.hip"), "-o", path] + [f"-D{d}" for d in defines]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def kernels(lines):
    """name -> instruction lines (labels kept) of every k_gemm_ar kernel"""
    out, name, body = {}, None, []
    for ln in lines:
        m = re.match(r"^(_ZN2gs9k_gemm_ar\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        s = ln.strip()
        if s.startswith(".Lfunc_end"):   # (a kernel may hold several s_endpgm: early exits come first)
            out[name] = body
            name = None
            continue
        if not s or s.startswith(";") or s.startswith("."):
            if s.startswith(".LBB"):
                body.append(s.split(":")[0] + ":")
            continue
        body.append(s.split(";")[0].strip())
    return out


def check(name, body):
    problems = []
    # loop = from the label that a backward s_cbranch targets to that branch
    labels = {ln[:-1]: i for i, ln in enumerate(body) if ln.endswith(":")}
    loop, loop_mf = None, 0
    for i, ln in enumerate(body):
        m = re.match(r"s_c?branch\w* (\.LBB\w+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            cand = (labels[m.group(1)], i)
            seg = body[cand[0]:cand[1]]
            # the main loop: three stages = three barriers per trip (other backward branches: the jump to the early
            # exit block, the epilogue's store loops); of several, the one with the most MFMAs
            if seg.count("s_barrier") == 3:
                mf = sum("mfma" in x for x in seg)
                if loop is None or mf > loop_mf:
                    loop, loop_mf = cand, mf
    if loop is None:
        return [f"{name}: no loop found"], {}
    # (the code behind the loop -- conditional tail stages, drain, epilogue -- is laid out in blocks whose file order is
    #  not their execution order: it is covered by the GPU parity tests with K of 1, 2 and 40 stages, not walked here)
    order = list(range(0, loop[1] + 1)) + list(range(loop[0], loop[1] + 1))
    queue = []          # outstanding vector-memory operations, oldest first: set of destination registers (may be empty)
    per_stage, stage = [], []
    in_loop_pass = 0
    for pos, i in enumerate(order):
        ln = body[i]
        if ln.endswith(":"):
            continue
        inside = loop[0] <= i <= loop[1]
        op = ln.split()[0]
        m = re.match(r"s_waitcnt (.*)", ln)
        if m:
            vm = re.search(r"vmcnt\((\d+)\)", m.group(1))
            if vm:
                n = int(vm.group(1))
                if n == 0 and inside:
                    problems.append(f"{name}: s_waitcnt vmcnt(0) inside the loop (line {i})")
                while len(queue) > n:
                    queue.pop(0)
            continue
        owned = set().union(*queue) if queue else set()
        touched = regs_of(ln.split(None, 1)[1]) if " " in ln else set()
        if op.startswith("global_load_lds"):
            queue.append(set())
            stage.append("DMA")
            continue
        if op.startswith("global_load") or op.startswith("buffer_load"):
            args = ln.split(None, 1)[1].split(",")
            dst, addr = regs_of(args[0]), regs_of(",".join(args[1:]))
            if addr & owned:
                problems.append(f"{name}: address of `{ln}` reads registers still owned by a load (line {i})")
            queue.append(dst)
            stage.append("LD")
            continue
        if op.startswith("global_store") or op.startswith("buffer_store"):
            if touched & owned:
                problems.append(f"{name}: `{ln}` reads registers still owned by a load (line {i})")
            queue.append(set())
            continue
        if touched & owned:
            problems.append(f"{name}: `{ln}` touches v{sorted(touched & owned)[:4]} while a load owns them (line {i})")
        if op == "s_barrier" and inside:
            per_stage.append(stage)
            stage = []
    return problems, {"stages": per_stage[:6], "loop_lines": loop[1] - loop[0] + 1}


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    defines = [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--define"]   # e.g. --define GS_AR_STAMPS
    path = keep or os.path.join(tempfile.mkdtemp(), "gemm_ar.s")
    compile_isa(path, defines)
    ks = kernels(open(path).read().splitlines())
    if not ks:
        print("no k_gemm_ar kernel in the assembly")
        return 1
    bad = 0
    for name, body in sorted(ks.items()):
        problems, info = check(name, body)
        tag = re.search(r"k_gemm_arILi(\d+)ELi(\d+)E", name)
        label = f"k_gemm_ar<{tag.group(1)} waves, {tag.group(2)} column tiles>" if tag else name
        stages = ["".join("D" if x == "DMA" else "L" for x in st) for st in info.get("stages", [])]
        print(f"{label}: loop of {info.get('loop_lines')} lines, vector-memory operations per stage {stages}: "
              f"{'clean' if not problems else str(len(problems)) + ' problem(s)'}")
        for p in problems[:10]:
            print("   ", p)
        bad += len(problems)
        for st in stages[1:]:   # (the first entry includes the prologue's requests)
            if not re.fullmatch(r"D+L{4}", st):
                print(f"    {label}: stage order {st} is not <pieces><4 loads>")
                bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
