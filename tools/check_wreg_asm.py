#!/usr/bin/env python3
"""Tripwire for the hand-counted weight loads and LDS fragment reads of gemm_wreg.hip (cdna_hip_programming.md section 5.7 item 1: the compiler does not
know an inline-asm load's destination is in flight, so a spill, copy or re-use of that register before the matching wait
would read garbage).  Compiles the file with -save-temps and, per kernel, collects every VGPR an asm global_load writes and
flags any COMPILER instruction that copies or spills one of them: v_mov_b32 / v_mov_b64 / v_accvgpr_write reading it, any
scratch_* naming it.  (MFMAs read them legitimately, behind the s_waitcnt statements that name them.)  Exit code 1 on a hit."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (source file, kernel name, how to read the template arguments out of the mangled name)
TARGETS = [
    ("gemm_wreg.hip", "gemm_wreg_kernel", r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)", "MB={0} WNB={1} NSTAGE={2} D={3} Q={4} L={5} ABL={6}", 6),
    ("gemm_w8a16.hip", "gemm_w8a16_kernel", r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)", "w8a16 MB={0} WNB={1} NSTAGE={2} D={3} L={4} ABL={5}", 5),
]


# the weight ring's loads: global_load_* until round 5, buffer_load_* (scalar k-step offset, immediate block offset) since round 6; gemm_w8a16.hip keeps global loads
RING_LOADS = ("global_load_dwordx4", "global_load_dwordx2", "buffer_load_dwordx4", "buffer_load_dwordx2")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main():
    bad = 0
    only = set(sys.argv[1:])                                  # optional: source file names to restrict the check to
    for fname, kname, tag_re, tag_fmt, abl_idx in TARGETS:
        if only and fname not in only:
            continue
        bad += check(fname, kname, tag_re, tag_fmt, abl_idx)
    return 1 if bad else 0


def check(fname, kname, tag_re, tag_fmt, abl_idx):
    src = os.path.join(ROOT, "mixq_amd", "csrc", fname)
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-save-temps", "-o", "x.o"], cwd=td,
                              stderr=subprocess.DEVNULL)
        text = open(os.path.join(td, fname.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    bad = 0
    for m in re.finditer(r"^(_ZN\S*" + kname + r"\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = body.split("\n")
        loaded, in_asm = set(), False
        for ln in lines:
            t = ln.strip()
            if t.startswith(";;#ASMSTART"): in_asm = True
            elif t.startswith(";;#ASMEND"): in_asm = False
            elif in_asm and t.startswith(RING_LOADS):   # (dwordx2: the 8-byte pieces of the FP6 form's fragments)
                loaded |= regs(t.split()[1].rstrip(","))
        # the k loop and its tail: from the first hand-placed wait to the last one
        first = last = None
        in_asm = seen_load = False
        for i, ln in enumerate(lines):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"): in_asm = True
            elif t.startswith(";;#ASMEND"): in_asm = False
            elif in_asm and t.startswith(RING_LOADS): seen_load = True
            elif in_asm and t.startswith("s_waitcnt vmcnt") and seen_load:       # (the loader wave's own waits come before any ring load)
                if first is None: first = i            # (the prologue gives every slot a defined value BEFORE its first load: harmless copies)
                if any("sched_barrier" in x for x in lines[i:i + 4]): last = i    # a ring wait (the trace drain at the very end is not)
        hits, in_asm, in_loop = [], False, False
        for i, ln in enumerate(lines):
            if re.match(r"^\.LBB\d+_\d+:", ln) or "Loop Header" in ln:
                # basic blocks the compiler marks as part of a loop: the unrolled k loop and its guarded tail.  (Blocks outside
                # loops re-use the same physical registers for other values on other paths; text alone cannot tell those apart.)
                in_loop = ("Loop Header" in ln) or ("in Loop:" in ln) or (in_loop and "=>This" in ln)
            if "Loop Header" in ln: in_loop = True
            if first is None or last is None or i < first or i > last or not in_loop:
                t = ln.strip()
                if t.startswith(";;#ASMSTART"): in_asm = True
                elif t.startswith(";;#ASMEND"): in_asm = False
                continue
            t = ln.strip()
            if t.startswith(";;#ASMSTART"): in_asm = True; continue
            if t.startswith(";;#ASMEND"): in_asm = False; continue
            if in_asm or not t or t.startswith((";", ".")): continue
            op = t.split()[0]
            toks = [x.strip(",") for x in t.split()[1:]]
            if op.startswith("scratch_"):
                if any(regs(x) & loaded for x in toks): hits.append((i, t))
            elif op.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr_write")):
                if any(regs(x) & loaded for x in toks[1:]): hits.append((i, t))          # a SOURCE in the ring
        # activation fragments (asm ds_read_b128 / ds_read_b64 + hand-counted lgkmcnt; gemm_wreg.hip since round 6, gemm_w8a16.hip before): LDS returns in order, so simulate the queue of outstanding
        # destination registers through the kernel text and flag any compiler copy / spill that READS a register still in it
        pending, in_asm, lds_hits, prev_op = [], False, [], ""
        for i, ln in enumerate(lines):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"): in_asm = True; continue
            if t.startswith(";;#ASMEND"): in_asm = False; continue
            if re.match(r"^\.LBB\d+_\d+:", t) and prev_op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                pending = []                                   # a block nothing falls into: the queue of the text above it is not its state
            if not t or t.startswith((";", ".")): continue
            op = t.split()[0]
            if not in_asm: prev_op = op
            toks = [x.strip(",") for x in t.split()[1:]]
            if in_asm:
                if op in ("ds_read_b128", "ds_read_b64"): pending.append(regs(toks[0]))   # (the FP6 forms read a fragment as 16 + 8 bytes: two queue entries)
                elif op == "s_waitcnt":
                    m2 = re.search(r"lgkmcnt\((\d+)\)", t)
                    if m2: pending = pending[len(pending) - int(m2.group(1)):] if int(m2.group(1)) else []
                continue
            if op == "s_waitcnt" and "lgkmcnt(0)" in t: pending = []
            inflight = set().union(*pending) if pending else set()
            if not inflight: continue
            if op.startswith("scratch_") and any(regs(x) & inflight for x in toks): lds_hits.append((i, t))
            elif op.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr_write")) and any(regs(x) & inflight for x in toks[1:]): lds_hits.append((i, t))
        hits += lds_hits
        # round 3: scratch accesses between the first ring request and the last ring wait are extra operations in the SAME in-order
        # queue the hand-counted vmcnt waits count.  They can only make a counted wait retire MORE than intended (loads retire in
        # order), so they are not an error by themselves - the error is a spill that READS a ring register in flight, flagged above -
        # but they are listed: a kernel that spills inside its k loop is one register away from that (the 256-row experiment of
        # DESIGN.md 6.00 went wrong exactly there).
        notes = []
        first_req = next((i for i, ln in enumerate(lines) if ln.strip().startswith(("global_load_dwordx4", "global_load_lds_dwordx4"))), None)
        if first_req is not None and last is not None:
            for i in range(first_req, last + 1):
                t = lines[i].strip()
                if t.startswith("scratch_"): notes.append((i, t))
        sc = re.search(r"; ScratchSize: (\d+)", body)
        tag = re.search(tag_re, name).groups()
        abl = tag[abl_idx] != "0"
        print(f"{tag_fmt.format(*tag)}: ring registers {len(loaded)}, in-flight LDS copies {len(lds_hits)}, "
              f"suspicious {len(hits)}" + ("  (ablation build: ignored)" if abl and hits else "") +
              (f"  [note: {len(notes)} scratch access(es) inside the counted region]" if notes else ""))
        for i, t in hits[:6]:
            print("     line", i, t)
        if hits and not abl:
            bad += 1
    return bad


if __name__ == "__main__":
    sys.exit(main())
