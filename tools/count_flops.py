#!/usr/bin/env python3
"""Counted floating-point work of the path, for the secondary (fp64 VALU) roofline of bench.py (last rows).

The CPU restatement (oracle/*.c) is compiled to LLVM IR (clang -O2, scalar, -ffp-contract=off like the reference build), every
basic block gets a call that adds its number of double-precision adds / multiplies / divisions and libm calls to counters, and
a sample of a bench workload runs through the instrumented library on one thread.  What comes out is the arithmetic the
reference's ALGORITHM executes per HRU-step -- its Brent iterations, Gauss-Seidel sweeps and all -- not what any particular
implementation spends; it plays the role for the VALU roofline that the algorithmic bytes play for the HBM roofline.

    python tools/count_flops.py [--config cfg3] [--ncell 64] [--steps 24]   ->  profiles/flops_<config>.json

TEST / MEASUREMENT INFRASTRUCTURE: builds oracle/_cnt/ (git-ignored); the product never loads it.
"""
import argparse, ctypes, json, os, re, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLANG = "/opt/rocm/lib/llvm/bin/clang"
OUT = os.path.join(ROOT, "oracle", "_cnt")
SRCS = ["orc_base.c", "orc_snow.c", "orc_surface.c", "orc_glacier.c", "orc_driver.c", "orc_putdata.c", "orc_blowing.c"]
NAMES = ["add_sub", "mul", "div", "exp", "log", "pow", "sqrt", "other_libm"]
LIBM = {"exp": 3, "exp2": 3, "log": 4, "log10": 4, "log2": 4, "pow": 5, "sqrt": 6, "llvm.sqrt.f64": 6, "cos": 7, "sin": 7, "tan": 7, "atan": 7,
        "cbrt": 7, "llvm.pow.f64": 5, "llvm.exp.f64": 3, "llvm.log.f64": 4, "llvm.log10.f64": 4, "llvm.cos.f64": 7, "llvm.exp2.f64": 3}
COUNTER_C = r"""
#include <string.h>
static __thread long long cnt[8];
void orc_cnt_add(long long a, long long m, long long d, long long e, long long l, long long p, long long s, long long o) {
  cnt[0] += a; cnt[1] += m; cnt[2] += d; cnt[3] += e; cnt[4] += l; cnt[5] += p; cnt[6] += s; cnt[7] += o;
}
void vicorc_cnt_read(long long *out, int reset) { memcpy(out, cnt, sizeof(cnt)); if (reset) memset(cnt, 0, sizeof(cnt)); }
"""


def lanes(line):
    m = re.search(r"= f(?:add|sub|mul|div)(?: [a-z]+)* <(\d+) x double>", line)
    return int(m.group(1)) if m else 1


def instrument(ll):
    """Insert a counting call after the phi nodes of every basic block with floating-point work."""
    out, block, in_func = [], [], False

    def flush():
        if not block:
            return
        c = [0] * 8
        for ln in block:
            if re.search(r"= f(add|sub)\b.*double", ln): c[0] += lanes(ln)
            elif re.search(r"= fmul\b.*double", ln): c[1] += lanes(ln)
            elif re.search(r"= fdiv\b.*double", ln): c[2] += lanes(ln)
            elif "@llvm.fmuladd.f64" in ln: c[0] += 1; c[1] += 1
            else:
                m = re.search(r"call .*double @([A-Za-z0-9_.]+)\(", ln)
                if m and m.group(1) in LIBM: c[LIBM[m.group(1)]] += 1
        if any(c):
            k = 0
            while k < len(block) and (block[k].rstrip().endswith(":") or re.match(r"^\S+:", block[k]) or " = phi " in block[k]):
                k += 1
            block.insert(k, "  call void @orc_cnt_add(i64 %d, i64 %d, i64 %d, i64 %d, i64 %d, i64 %d, i64 %d, i64 %d)\n" % tuple(c))
        out.extend(block)
        block.clear()
    for ln in ll.splitlines(keepends=True):
        if ln.startswith("define "):
            in_func = True
            out.append(ln)
            continue
        if in_func and ln.startswith("}"):
            flush(); in_func = False
            out.append(ln)
            continue
        if not in_func:
            out.append(ln)
            continue
        if re.match(r"^[A-Za-z0-9_.$-]+:", ln):        # a label starts a new block
            flush()
        block.append(ln)
    out.append("\ndeclare void @orc_cnt_add(i64, i64, i64, i64, i64, i64, i64, i64)\n")
    return "".join(out)


def build():
    os.makedirs(OUT, exist_ok=True)
    lls = []
    for s in SRCS:
        ll = os.path.join(OUT, s[:-2] + ".ll")
        subprocess.check_call([CLANG, "-O2", "-ffp-contract=off", "-fno-vectorize", "-fno-slp-vectorize", "-fPIC", "-std=gnu11", "-w", "-S", "-emit-llvm",
                               "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "oracle", s), "-o", ll])
        with open(ll) as f:
            txt = instrument(f.read())
        with open(ll, "w") as f:
            f.write(txt)
        lls.append(ll)
    cc = os.path.join(OUT, "counter.c")
    with open(cc, "w") as f:
        f.write(COUNTER_C)
    lib = os.path.join(OUT, "libvicoracle_counted.so")
    subprocess.check_call([CLANG, "-O2", "-fPIC", "-shared", "-w", "-o", lib] + lls + [cc, "-lm"])
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--ncell", type=int, default=64)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    args = ap.parse_args()
    lib_path = build()
    import bench
    from vic_amd import domain, init_state
    from oracle import pyref
    cfg = bench.config(args.config)
    opt = cfg["opt"]
    d = domain.make_domain(args.ncell, opt, ntile=cfg["ntile"], glacier_top_band=cfg.get("glacier", False))
    f, sf, dmy = domain.make_forcing(d, 0, args.warmup + args.steps, start_doy=cfg["start_doy"])
    sd0, si0 = init_state.initial_state(d, f[0])
    pyref.oracle_lib_path = lambda: lib_path                      # the instrumented build instead of oracle/libvicoracle.so
    m = pyref.OracleModel(d)
    m.set_state(sd0, si0)
    rd = m.lib.vicorc_cnt_read; rd.restype = None; rd.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    c = (ctypes.c_longlong * 8)()
    for s in range(args.warmup):
        m.step(f[s], sf[s], dmy[s])
    rd(c, 1)
    for s in range(args.warmup, args.warmup + args.steps):
        m.step(f[s], sf[s], dmy[s])
    rd(c, 1)
    n = d.nhru * args.steps
    per = {k: c[i] / n for i, k in enumerate(NAMES)}
    # one flop per add/sub/mul; a division, square root or libm call is counted as ONE operation here and listed separately
    # (their cost in VALU instructions is implementation specific: ~10 for a division, ~30-100 for exp / log / pow)
    flops = per["add_sub"] + per["mul"] + per["div"] + per["sqrt"]
    libm = per["exp"] + per["log"] + per["pow"] + per["other_libm"]
    out = {"config": args.config, "sample": "%d cells x %d steps after %d warm-up steps (same hours of the day as bench.py's default run)" % (args.ncell, args.steps, args.warmup),
           "hru_per_cell": d.nhru // d.ncell, "per_hru_step": per, "fp64_ops_per_hru_step": flops, "libm_calls_per_hru_step": libm,
           "fp64_ops_per_cell_step": flops * (d.nhru // d.ncell), "libm_calls_per_cell_step": libm * (d.nhru // d.ncell),
           "method": "tools/count_flops.py: oracle/*.c -> LLVM IR (clang -O2, scalar, fp-contract off), per-basic-block counters, one thread"}
    path = os.path.join(ROOT, "profiles", "flops_%s.json" % args.config)
    with open(path, "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
