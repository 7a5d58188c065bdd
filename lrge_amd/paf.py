"""PAF emission: turns the chain records of `Index.chains` + the per-query seed
statistics of `Index.paf_stats` into the lines liblrge writes to overlaps.paf
(liblrge/src/minimap2/mapping.rs:10-54 field order, :81-177 serialisers; aligner.rs:244-291 field sources).

All integer fields come from the device.  dv is finished here on the host, in double with libm's pow,
exactly as minimap2's mm_est_err does (mm2:esterr.c): the device supplies n_match (= cnt: every chain
anchor is itself an entry of mini_pos), the number of kept seeds spanned by the chain, and avg_k.
"""
import math

import numpy as np

F32_EPS = float(np.finfo(np.float32).eps)


def chain_dv(chain, qlen, tlen, sum_span, n_kept):
    """mm_est_err for one chain -> f32."""
    if n_kept == 0:
        return np.float32(-1.0)
    avg_k = np.float32(np.float32(sum_span) / np.float32(n_kept))       # (float)sum_k / n
    n_match = int(chain["cnt"])
    n_tot = int(chain["n_seeds"])
    qs, rs, re = int(chain["qs"]), int(chain["rs"]), int(chain["re"])
    if np.float32(qs) > avg_k and np.float32(rs) > avg_k:
        n_tot += 1
    if np.float32(qlen - qs) > avg_k and np.float32(tlen - re) > avg_k:
        n_tot += 1
    if n_match >= n_tot:
        return np.float32(0.0)
    return np.float32(1.0 - math.pow(n_match / n_tot, 1.0 / float(avg_k)))


def format_dv(dv):
    """mapping.rs:136-147: `0` below f32::EPSILON, else 4 decimals."""
    return "0" if float(dv) < F32_EPS else "%.4f" % float(dv)


def paf_lines(chains, q_names, q_lens, t_names, t_lens, rep_len, sum_span, n_kept):
    """One PAF line (str, no newline) per chain.  Order is undefined in the reference (rayon workers write
    under a mutex as they finish): compare as a multiset."""
    out = []
    for c in chains:
        q, t = int(c["query"]), int(c["target"])
        dv = chain_dv(c, int(q_lens[q]), int(t_lens[t]), int(sum_span[q]), int(n_kept[q]))
        qn = q_names[q].decode() if isinstance(q_names[q], bytes) else q_names[q]
        tn = t_names[t].decode() if isinstance(t_names[t], bytes) else t_names[t]
        out.append("\t".join([qn, str(int(q_lens[q])), str(int(c["qs"])), str(int(c["qe"])), "-" if c["rev"] else "+",
                              tn, str(int(t_lens[t])), str(int(c["rs"])), str(int(c["re"])), str(int(c["mlen"])),
                              str(int(c["blen"])), "0", "tp:A:S", "cm:i:%d" % int(c["cnt"]), "s1:i:%d" % int(c["score"]),
                              "dv:f:" + format_dv(dv), "rl:i:%d" % int(rep_len[q])]))
    return out
