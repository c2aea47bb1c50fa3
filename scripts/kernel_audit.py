#!/usr/bin/env python3
"""Static audit of the device code inside libhmx.so: registers, spills, scratch, code size and a few instruction
counts per kernel -- the things that decided k_round's speed in round 2: a spilled register is
reloaded behind `s_waitcnt vmcnt(0)` (drains every load in flight), a pointer that lost its address space turns LDS
reads into flat loads (which also count on vmcnt).

    python scripts/kernel_audit.py [libhmx.so] [name filter]

Needs only the LLVM tools of the ROCm image (no GPU).  `audit()` is what tests/test_kernel_audit.py calls."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("HMX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def tools_available():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf", "llvm-objdump"))


def _run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def extract_code_objects(lib, workdir):
    """one gfx950 code object per translation unit: .hip_fatbin is a sequence of offload bundles"""
    fat = os.path.join(workdir, "fat.bin")
    _run(os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(workdir, "unused.so"))
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    cos = []
    for i, st in enumerate(starts):
        part = os.path.join(workdir, f"bundle{i}.bin")
        with open(part, "wb") as f:
            f.write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(workdir, f"gfx950_{i}.co")
        _run(os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={part}", f"--output={co}")
        if os.path.getsize(co) > 0:
            cos.append(co)
    return cos


def kernel_metadata(co):
    """{mangled name: {vgpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size}}"""
    notes = _run(os.path.join(LLVM, "llvm-readelf"), "--notes", co)
    out, cur = {}, None
    keys = ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
            "group_segment_fixed_size", "agpr_count")
    # a kernel's map lists .name among its keys (after .args, whose entries have .name too: those are indented deeper)
    for block in re.split(r"\n  - \.", "\n" + notes):
        m = re.search(r"^    \.name:\s+(\S+)$", block, re.M)
        if not m or ".symbol:" not in block:
            continue
        cur = {k: int(v) for k, v in re.findall(r"^    \.(\w+):\s+(\d+)$", block, re.M) if k in keys}
        out[m.group(1)] = cur
    return out


def instruction_counts(co, names):
    """per kernel: code bytes and counts of a few instruction classes (from the disassembly)"""
    dis = _run(os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co)
    res = {}
    for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.M | re.S):
        name, body = m.group(1), m.group(2)
        if name not in names:
            continue
        lines = [l for l in body.split("\n") if l.strip()]
        addr = [int(a, 16) for a in re.findall(r"//\s*([0-9A-Fa-f]{6,16}):", body)]
        res[name] = {
            "instructions": len(lines),
            "code_bytes": (max(addr) - min(addr) + 8) if addr else 0,
            "mfma": sum("v_mfma" in l for l in lines),
            "flat": sum(re.search(r"\bflat_(load|store|atomic)", l) is not None for l in lines),
            "scratch": sum("scratch_" in l for l in lines),
            "lds_dma": sum("global_load_lds" in l for l in lines),
            "vmcnt0": sum(re.search(r"s_waitcnt.*vmcnt\(0\)", l) is not None for l in lines),
            "barriers": sum("s_barrier" in l for l in lines),
        }
    return res


# ------------------------------------------------------------------------------------------------------------------
# Loads in flight: nothing may touch the destination registers of a vector-memory load before a wait has retired it.
# The compiler keeps that rule for the loads it issues; a load issued from INLINE ASSEMBLY (k_assign_wide2's Z pieces)
# is invisible to its wait insertion, and rounds 2-3 shipped a kernel whose `v_mov` copies of such registers sat in
# front of the hand-counted s_waitcnt ("the wait that named its registers").  This pass replays the
# gfx9 rule on the BUILT code of every kernel: vector-memory operations retire in order, `s_waitcnt vmcnt(N)` leaves at
# most the N youngest outstanding (the counter saturates at 63); every control-flow path is followed (states = queues
# of destination register sets, explored to a fixed point).
# ------------------------------------------------------------------------------------------------------------------
_VMEM = re.compile(r"^(global|flat|buffer|scratch|tbuffer)_(load|store|atomic)")
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _vregs(text):
    out = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _parse_function(body):
    """[(mnemonic, operand text)], {label: index}"""
    ins, labels = [], {}
    for line in body.split("\n"):
        line = line.split("//")[0].strip()
        if not line:
            continue
        m = re.match(r"^[0-9a-f]+ <(L\d+)>:$", line)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        parts = line.split(None, 1)
        ins.append((parts[0], parts[1] if len(parts) > 1 else ""))
    return ins, labels


def inflight_hazards_in(body):
    """hazards of one kernel's disassembly (llvm-objdump -d --symbolize-operands): [(text, registers, instruction index)].
    State at an instruction = {register with a load in flight: number of YOUNGER vector-memory operations}; an operation
    ages every entry by one (64 outstanding cannot exist: the counter saturates), `vmcnt(N)` retires the entries with at
    least N younger operations, a join keeps the smaller count -- a monotone data-flow problem, solved to its fixed point.
    A load whose own DESTINATION is still in flight is not reported (two loads into one register retire in order and the
    first value is dead: that is what the two mutually exclusive copies of k_round's request code look like to a
    path-insensitive walk); anything else that names such a register is."""
    ins, labels = _parse_function(body)
    n = len(ins)
    state = [None] * (n + 1)
    state[0] = {}
    hazards = {}
    work = [0]

    def merge(pc, st):
        if pc > n:
            return
        cur = state[pc]
        if cur is None:
            state[pc] = dict(st)
            work.append(pc)
            return
        changed = False
        for r, c in st.items():
            if r not in cur or c < cur[r]:
                cur[r] = c
                changed = True
        if changed:
            work.append(pc)

    while work:
        pc = work.pop()
        if pc >= n:
            continue
        st = state[pc]
        mn, ops = ins[pc]
        out = st
        if mn == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ops)
            if m:
                lim = int(m.group(1))
                out = {r: c for r, c in st.items() if c < lim}
        else:
            regs = _vregs(ops)
            vmem = _VMEM.match(mn) is not None
            dest = set()
            if vmem:
                returning = "_load" in mn or ("_atomic" in mn and re.search(r"\b(sc0|glc)\b", ops) is not None)
                if returning and not mn.startswith("global_load_lds"):
                    dest = _vregs(ops.split(",")[0])
            touched = (regs - dest if vmem else regs) & set(st)
            if touched:
                hazards[pc] = (f"{mn} {ops}", sorted(touched), pc)
            if vmem:
                out = {r: c + 1 for r, c in st.items() if c + 1 < 64}
                for r in dest:
                    out[r] = 0
        if mn == "s_endpgm":
            continue
        if mn == "s_branch":
            merge(labels[ops.strip()], out)
            continue
        if mn.startswith("s_cbranch"):
            merge(labels[ops.strip()], out)
        merge(pc + 1, out)
    return [hazards[k] for k in sorted(hazards)]


def inflight_hazards(lib=None, name_filter=""):
    """{kernel: [(text, registers, index)]} for every kernel of the library"""
    lib = lib or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "harmonypy_amd", "libhmx.so")
    res = {}
    with tempfile.TemporaryDirectory() as wd:
        for co in extract_code_objects(lib, wd):
            names = set(kernel_metadata(co))
            dis = _run(os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--symbolize-operands", co)
            for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^[0-9a-f]+ <[^L][^>]*>:|\Z)", dis, re.M | re.S):
                if m.group(1) in names and name_filter in m.group(1):
                    res[m.group(1)] = inflight_hazards_in(m.group(2))
    return res


def demangle(names):
    try:
        out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return dict(zip(names, out.split("\n")))
    except Exception:
        return {n: n for n in names}


def audit(lib=None, name_filter=""):
    """[{name, demangled, vgpr_count, ..., mfma, flat, scratch, ...}] for the kernels whose demangled name contains name_filter"""
    lib = lib or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "harmonypy_amd", "libhmx.so")
    with tempfile.TemporaryDirectory() as wd:
        meta, ins = {}, {}
        for co in extract_code_objects(lib, wd):
            m = kernel_metadata(co)
            meta.update(m)
            ins.update(instruction_counts(co, {n for n in m if name_filter in n or name_filter in demangle([n])[n]}))
        dm = demangle(list(meta))
        pick = [n for n in meta if name_filter in dm[n] or name_filter in n]
    return [dict(name=n, demangled=dm[n], **meta[n], **ins.get(n, {})) for n in sorted(pick, key=lambda x: dm[x])]


if __name__ == "__main__":
    if not tools_available():
        sys.exit(f"LLVM tools not found under {LLVM}")
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else None
    flt = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].endswith(".so") and sys.argv[-1] != "--inflight" else ""
    if "--inflight" in sys.argv:
        hz = inflight_hazards(lib, flt if flt != "--inflight" else "")
        bad = {k: v for k, v in hz.items() if v}
        print(f"{len(hz)} kernels followed, {len(bad)} with a register touched while a load into it was in flight")
        for k, v in bad.items():
            print(" ", demangle([k])[k], v[:6])
        sys.exit(1 if bad else 0)
    rows = audit(lib, flt)
    print(f"{'kernel':58s} {'vgpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratchB':>8s} {'KB':>6s} {'mfma':>5s} {'flat':>5s} {'scr':>4s} {'dma':>4s} {'vm0':>4s} {'bar':>4s}")
    for r in rows:
        print(f"{r['demangled'][:58]:58s} {r.get('vgpr_count', 0):5d} {r.get('sgpr_count', 0):5d} {r.get('vgpr_spill_count', 0):6d} "
              f"{r.get('sgpr_spill_count', 0):6d} {r.get('private_segment_fixed_size', 0):8d} {r.get('code_bytes', 0) / 1024:6.1f} "
              f"{r.get('mfma', 0):5d} {r.get('flat', 0):5d} {r.get('scratch', 0):4d} {r.get('lds_dma', 0):4d} {r.get('vmcnt0', 0):4d} {r.get('barriers', 0):4d}")
