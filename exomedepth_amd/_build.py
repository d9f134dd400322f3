"""Build libedcore.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libedcore.so")
SOURCES = ["edcore.hip"]
def _deps():
    """every file the library is built from: all of csrc/ (sources, .inc, headers -- generated tables included) + the C-ABI header"""
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".inc", ".hpp", ".h")))
    return names + ["../../include/exomedepth_amd.h"]


DEPS = _deps()
# -ffp-contract=off is part of the numerical contract (see csrc/ed_pmath.h): no implicit fma.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fvisibility=hidden"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def csrc_sha16():
    """Fingerprint of the kernel sources (the files libedcore.so is built from).  rocprofv3 summaries under profiles/ are
    stamped with it (tools/profile_round.sh); bench.py only quotes counter-derived figures from a profile whose stamp
    equals the tree's, so a changed kernel can never be described by a stale profile."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS):
        p = os.path.join(CSRC, d)
        if os.path.exists(p):
            h.update(d.encode()); h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for d in DEPS:
        p = os.path.join(CSRC, d)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(" ".join(cmd))
        print(r.stdout)
        print(r.stderr)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed building libedcore.so")
    return LIB


# Diagnostic variants of the library (same sources, different code generation); never loaded by the package itself.
VARIANTS = {"coldinline": ["-DED_COLD_INLINE"],
            # the library with its experiment knobs (environment variables ED_TAB_REACH, ED_TAB_TW, ED_VIT_WAVES, ED_SM_NSPLIT, ED_COHORT_*) enabled
            "knobs": ["-DED_EXPERIMENT_KNOBS"],
            # k_viterbi_sm without its raised wave priority (round 5 A/B: tools/ab.sh vitprio0)
            "vitprio0": ["-DED_VITSM_PRIO=0"], "prepprio3": ["-DED_PREP_PRIO=3"], "prio_prep_over_vit": ["-DED_VITSM_PRIO=0", "-DED_PREP_PRIO=3"], "vitdepth1": ["-DED_VITSM_DEPTH=1"], "fitpre4": ["-DED_FIT_PRE=4"], "tabbuild256": ["-DED_TAB_BUILD_THREADS=256"],
            # (the "xu16" timing build -- k_emit_tab_sm reading its counts as if 16 bits wide, profiles/r05_u16_experiment.txt -- became the real thing:
            #  ed_batch_set_counts_bits(batch, 16))
            # timing experiments on k_emit_tab_sm (wrong results by construction): without its stores / LDS look-ups / global look-ups
            "xnostore": ["-DED_SM_X_NOSTORE"], "xnolds": ["-DED_SM_X_NOLDS"], "xnoglobal": ["-DED_SM_X_NOGLOBAL"],
            "xnoldsglobal": ["-DED_SM_X_NOLDS", "-DED_SM_X_NOGLOBAL"],
            "tabper8": ["-DED_TAB_PER=8"],
            # round 1's Horner step (coefficient as an "s" asm operand): contains the VALU-write-SGPR -> VALU-read hazard
            # (tools/isa_hazard_scan.py); built only to demonstrate it on hardware next to the fixed library
            "sgprasm": ["-DED_PM_FMA_K_SGPR_OPERAND"],
            # k_fit_hist of the shallow geometry with 4 samples per workgroup: half the LDS, emission workgroups fit beside it
            "fitlight": ["-DED_HG8_WG=4"],
            # every automatic variable starts from a bit pattern (0xAA...) instead of whatever the register or stack slot held: a read of an uninitialised
            # variable shows up as a wrong (and reproducible) result instead of a run-to-run difference -- host and device code
            "autoinit": ["-ftrivial-auto-var-init=pattern"],
            # k_fit_hnewton's per-cell path with the short digamma series (irreproducible fits), alone and with pattern-initialised variables
            "hnshort": ["-DED_HN_SHORT_SERIES"], "hnshortinit": ["-DED_HN_SHORT_SERIES", "-ftrivial-auto-var-init=pattern"],
            "hnshortplain": ["-DED_HN_SHORT_SERIES", "-DED_FIT_PLAIN_FMA"], "fitplain": ["-DED_FIT_PLAIN_FMA"],
            # host code under the sanitizers (device code is left alone: -fno-gpu-sanitize); tools/sanitize.sh
            # (no -shared-libsan: the runtime is whatever tools/sanitize.sh preloads -- gcc's stock libasan / libtsan; ROCm's own
            # ASan runtime intercepts hsa_amd_memory_pool_allocate for DEVICE instrumentation and fails on a plain process)
            "asan": ["-fsanitize=address,undefined", "-fno-sanitize=vptr,function", "-fno-gpu-sanitize", "-g", "-fno-omit-frame-pointer", "-Wl,--unresolved-symbols=ignore-all"],
            "tsan": ["-fsanitize=thread", "-fno-gpu-sanitize", "-g", "-fno-omit-frame-pointer", "-Wl,--unresolved-symbols=ignore-all"],
            # k_emit_tab_sm experiments: plain loads / stores; 512-thread workgroups with half the LDS (two per CU)
            "smplain": ["-DED_SM_NT=0"], "smntld": ["-DED_SM_NT=1"], "smntst": ["-DED_SM_NT=2"],
            "sm512": ["-DED_SM_THREADS=512", "-DED_SM_ENTRIES=3072"],
            # ... 8- and 12-wave workgroups with the whole table window (one per CU, registers left on every SIMD for a Viterbi / table-build wave)
            "nofence": ["-DED_X_NO_NULL_FENCE"], "sm512full": ["-DED_SM_THREADS=512"], "sm768full": ["-DED_SM_THREADS=768"],
            "smmask": ["-DED_SM_MASKED=1"], "sm6784": ["-DED_SM_ENTRIES=6784"], "smmask6784": ["-DED_SM_MASKED=1", "-DED_SM_ENTRIES=6784"], "sm4096": ["-DED_SM_ENTRIES=4096"], "sm3584": ["-DED_SM_ENTRIES=3584"], "sm5120": ["-DED_SM_ENTRIES=5120"]}


def variant_path(name):
    return os.path.join(HERE, "libedcore_%s.so" % name)


def build_variant(name, force=False, verbose=False):
    out = variant_path(name)
    if not force and os.path.exists(out) and not stale() and os.path.getmtime(out) >= os.path.getmtime(LIB):
        return out
    cmd = [hipcc()] + FLAGS + VARIANTS[name] + ["-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(" ".join(cmd)); print(r.stdout); print(r.stderr)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed building %s" % out)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
