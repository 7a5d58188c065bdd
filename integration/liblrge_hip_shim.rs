//! liblrge_hip_shim.rs -- the binding a liblrge maintainer drops into `liblrge/src/` (as `hip.rs`, `mod hip;` in
//! lib.rs, cargo feature `hip`, `println!("cargo:rustc-link-lib=dylib=lrge_hip")` in build.rs) to run the overlap hot
//! path of `TwoSetStrategy` / `AvaStrategy` on an MI355X through `liblrge_hip.so` (C ABI: include/lrge_hip.h).
//!
//! NOT compiled in this repository: the image has no Rust toolchain.  It is written against
//! liblrge 0.2's own items and nothing else:
//!   crate::{Result, error::LrgeError, io::iter_records}   (lib.rs:145, error.rs:6-33, io.rs:149-184)
//!   crate::estimate::Estimate                              (estimate.rs:21-78 -- the trait is untouched: the default
//!                                                           `estimate()` keeps calling `generate_estimates()`)
//! What changes in the reference: the bodies of `generate_estimates` (twoset.rs:587-606, ava.rs:369-381) call
//! `hip::twoset_estimates` / `hip::ava_estimates` below instead of `AlignerWrapper::new` + `align_reads*`; `split_fastq`
//! and `subsample_reads` stay as they are (the FASTA files they write are read back here with the crate's own
//! `iter_records`, exactly what `mm_idx_reader_open` and the producer thread of `align_reads` do today).

#![allow(non_camel_case_types, dead_code)]

use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::path::Path;
use std::ptr;

use crate::error::LrgeError;
use crate::io::iter_records;

// ------------------------------------------------------------------------------------------------------------------
// raw FFI (include/lrge_hip.h)
// ------------------------------------------------------------------------------------------------------------------
#[repr(C)] pub struct lrge_hip_ctx { _p: [u8; 0] }
#[repr(C)] pub struct lrge_hip_seqset { _p: [u8; 0] }
#[repr(C)] pub struct lrge_hip_index { _p: [u8; 0] }
#[repr(C)] pub struct lrge_hip_comm { _p: [u8; 0] }

#[repr(C)]
#[derive(Clone, Copy)]
pub struct lrge_hip_params {
    pub remove_internal: i32,
    pub max_overhang_ratio: f32,
}

/// One chain = one mm_reg1_t = one PafRecord (include/lrge_hip.h: lrge_hip_chain; aligner.rs:253-290).
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct lrge_hip_chain {
    pub query: u32,
    pub target: u32,
    pub rev: i32,
    pub score: i32,
    pub cnt: i32,
    pub qs: i32,
    pub qe: i32,
    pub rs: i32,
    pub re: i32,
    pub mlen: i32,
    pub blen: i32,
    pub n_seeds: i32,
}

pub const LRGE_PRESET_AVA_ONT: c_int = 0; // Preset::AvaOnt (preset.rs:26)
pub const LRGE_PRESET_AVA_PB: c_int = 1; //  Preset::AvaPb  (preset.rs:24)

extern "C" {
    fn lrge_hip_ctx_create(device: c_int, out: *mut *mut lrge_hip_ctx) -> c_int;
    fn lrge_hip_ctx_destroy(ctx: *mut lrge_hip_ctx);
    fn lrge_hip_last_error(ctx: *const lrge_hip_ctx) -> *const c_char;
    fn lrge_hip_seqset_upload(ctx: *mut lrge_hip_ctx, bases: *const c_char, offsets: *const u64, n: u32,
                              name_rank: *const u32, out: *mut *mut lrge_hip_seqset) -> c_int;
    fn lrge_hip_seqset_free(s: *mut lrge_hip_seqset);
    fn lrge_hip_seqset_presketch(ctx: *mut lrge_hip_ctx, s: *mut lrge_hip_seqset, preset: c_int) -> c_int;
    fn lrge_hip_index_build(ctx: *mut lrge_hip_ctx, targets: *const lrge_hip_seqset, preset: c_int,
                            out: *mut *mut lrge_hip_index) -> c_int;
    fn lrge_hip_index_build_for(ctx: *mut lrge_hip_ctx, targets: *const lrge_hip_seqset, preset: c_int,
                                streamed: *mut lrge_hip_seqset, comm: *mut lrge_hip_comm,
                                out: *mut *mut lrge_hip_index) -> c_int;
    fn lrge_hip_index_build_tsharded(ctx: *mut lrge_hip_ctx, target_shard: *const lrge_hip_seqset, preset: c_int, comm: *mut lrge_hip_comm,
                                     out: *mut *mut lrge_hip_index) -> c_int;
    fn lrge_hip_index_build_sharded(ctx: *mut lrge_hip_ctx, all_target_lens: *const u32, all_target_ranks: *const u32,
                                    n_targets: u32, target_shard: *const lrge_hip_seqset, shard_first: u32, preset: c_int,
                                    streamed: *mut lrge_hip_seqset, comm: *mut lrge_hip_comm,
                                    out: *mut *mut lrge_hip_index) -> c_int;
    fn lrge_hip_index_free(ix: *mut lrge_hip_index);
    fn lrge_hip_overlap_twoset(ctx: *mut lrge_hip_ctx, ix: *const lrge_hip_index, queries: *const lrge_hip_seqset,
                               p: *const lrge_hip_params, counts: *mut u32, has_mapping: *mut u32) -> c_int;
    fn lrge_hip_overlap_inverse(ctx: *mut lrge_hip_ctx, ix: *const lrge_hip_index, streamed: *const lrge_hip_seqset,
                                p: *const lrge_hip_params, counts: *mut u32) -> c_int;
    fn lrge_hip_overlap_ava(ctx: *mut lrge_hip_ctx, ix: *const lrge_hip_index, reads: *const lrge_hip_seqset,
                            p: *const lrge_hip_params, counts: *mut u32) -> c_int;
    fn lrge_hip_estimates(ctx: *mut lrge_hip_ctx, counts: *const u32, read_lens: *const u32, n: u32,
                          avg_target_len: f32, n_target_reads: u64, overlap_thresh: u32, out: *mut f32) -> c_int;
    // the PafRecord stream (overlaps.paf: twoset.rs:244-250,293, :410-416,487, ava.rs:220-226,273)
    fn lrge_hip_chains(ctx: *mut lrge_hip_ctx, ix: *const lrge_hip_index, queries: *const lrge_hip_seqset, dual: c_int,
                       out: *mut lrge_hip_chain, cap: u64, n_out: *mut u64) -> c_int;
    fn lrge_hip_paf_stats(ctx: *mut lrge_hip_ctx, ix: *const lrge_hip_index, queries: *const lrge_hip_seqset,
                          rep_len: *mut i32, sum_span: *mut u64, n_kept: *mut u32) -> c_int;
    fn lrge_hip_seqset_presketch_sharded(ctx: *mut lrge_hip_ctx, s: *mut lrge_hip_seqset, preset: c_int, comm: *mut lrge_hip_comm) -> c_int;
    // multi-GPU (one thread per GPU inside this process: liblrge already owns a rayon pool)
    fn lrge_hip_comm_local_group_create(world: c_int, group: *mut *mut c_void) -> c_int;
    fn lrge_hip_comm_local_group_destroy(group: *mut c_void);
    fn lrge_hip_comm_create_local(ctx: *mut lrge_hip_ctx, rank: c_int, group: *mut c_void,
                                  out: *mut *mut lrge_hip_comm) -> c_int;
    fn lrge_hip_comm_destroy(c: *mut lrge_hip_comm);
    fn lrge_hip_comm_abort(c: *mut lrge_hip_comm) -> c_int;
    fn lrge_hip_comm_allgather(c: *mut lrge_hip_comm, send: *const c_void, bytes: usize, recv: *mut c_void) -> c_int;
    fn lrge_hip_comm_allreduce_u32(c: *mut lrge_hip_comm, inout: *mut u32, n: usize) -> c_int;
}

// ------------------------------------------------------------------------------------------------------------------
// error mapping: LRGE_ERR_* -> LrgeError (lrge_hip.h lists the codes next to the variants of error.rs:6-33)
// ------------------------------------------------------------------------------------------------------------------
fn to_err(rc: c_int, ctx: *const lrge_hip_ctx) -> LrgeError {
    let msg = unsafe {
        let p = lrge_hip_last_error(ctx);
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    match rc {
        -1 => LrgeError::IoError(std::io::Error::new(std::io::ErrorKind::Other, msg)),
        -2 => LrgeError::FastqParseError(msg),
        -3 => LrgeError::TooManyReadsError(msg),
        -4 => LrgeError::TooFewReadsError(msg),
        -6 => LrgeError::MapError(msg),
        -7 => LrgeError::DuplicateReadIdentifier(msg),
        -8 => LrgeError::PafWriteError(msg),
        _ => LrgeError::ThreadError(msg), // -5 device / -9 invalid argument: the "threads / device" class
    }
}

macro_rules! check {
    ($ctx:expr, $call:expr) => {{
        let rc = unsafe { $call };
        if rc != 0 { return Err(to_err(rc, $ctx)); }
    }};
}

// ------------------------------------------------------------------------------------------------------------------
// owned handles
// ------------------------------------------------------------------------------------------------------------------
pub struct Ctx { h: *mut lrge_hip_ctx }
pub struct SeqSet<'c> { h: *mut lrge_hip_seqset, n: u32, _ctx: &'c Ctx }
pub struct Index<'c> { h: *mut lrge_hip_index, _ctx: &'c Ctx }

// A context serves one call at a time (lrge_hip.h); it may move between threads but is not shared.
unsafe impl Send for Ctx {}

impl Ctx {
    pub fn new(device: i32) -> crate::Result<Self> {
        let mut h = ptr::null_mut();
        check!(ptr::null(), lrge_hip_ctx_create(device as c_int, &mut h));
        Ok(Ctx { h })
    }

    pub fn upload<'c>(&'c self, reads: &ReadSet, ranks: &[u32]) -> crate::Result<SeqSet<'c>> {
        let mut h = ptr::null_mut();
        check!(self.h, lrge_hip_seqset_upload(self.h, reads.bases.as_ptr() as *const c_char, reads.offsets.as_ptr(),
                                              reads.len() as u32, ranks.as_ptr(), &mut h));
        Ok(SeqSet { h, n: reads.len() as u32, _ctx: self })
    }

    /// Optional hint: `s` is sketched beside the next index build (results unchanged).
    pub fn presketch(&self, s: &SeqSet, preset: c_int) -> crate::Result<()> {
        check!(self.h, lrge_hip_seqset_presketch(self.h, s.h, preset));
        Ok(())
    }

    /// AlignerWrapper::new(target_file, threads, preset, dual) -- aligner.rs:310-328.
    pub fn index<'c>(&'c self, targets: &SeqSet<'c>, preset: c_int) -> crate::Result<Index<'c>> {
        let mut h = ptr::null_mut();
        check!(self.h, lrge_hip_index_build(self.h, targets.h, preset, &mut h));
        Ok(Index { h, _ctx: self })
    }

    /// The index of one rank of a multi-GPU run: built for `streamed` only, statistics over all targets.
    pub fn index_for<'c>(&'c self, targets: &SeqSet<'c>, preset: c_int, streamed: &SeqSet<'c>,
                         comm: *mut lrge_hip_comm) -> crate::Result<Index<'c>> {
        let mut h = ptr::null_mut();
        check!(self.h, lrge_hip_index_build_for(self.h, targets.h, preset, streamed.h, comm, &mut h));
        Ok(Index { h, _ctx: self })
    }

    /// twoset.rs:286-317: (distinct-target counts, has_mapping) per query.
    pub fn overlap_twoset(&self, ix: &Index, queries: &SeqSet, p: &lrge_hip_params) -> crate::Result<(Vec<u32>, Vec<u32>)> {
        let n = queries.n as usize;
        let (mut counts, mut has) = (vec![0u32; n.max(1)], vec![0u32; n.max(1)]);
        check!(self.h, lrge_hip_overlap_twoset(self.h, ix.h, queries.h, p, counts.as_mut_ptr(), has.as_mut_ptr()));
        counts.truncate(n); has.truncate(n);
        Ok((counts, has))
    }

    /// twoset.rs:485-524: per INDEXED read, the number of streamed reads with a kept mapping onto it.
    pub fn overlap_inverse(&self, ix: &Index, n_indexed: usize, streamed: &SeqSet, p: &lrge_hip_params) -> crate::Result<Vec<u32>> {
        let mut counts = vec![0u32; n_indexed.max(1)];
        check!(self.h, lrge_hip_overlap_inverse(self.h, ix.h, streamed.h, p, counts.as_mut_ptr()));
        counts.truncate(n_indexed);
        Ok(counts)
    }

    /// ava.rs:271-306: symmetric overlap counts, one per read.
    pub fn overlap_ava(&self, ix: &Index, reads: &SeqSet, p: &lrge_hip_params) -> crate::Result<Vec<u32>> {
        let n = reads.n as usize;
        let mut counts = vec![0u32; n.max(1)];
        check!(self.h, lrge_hip_overlap_ava(self.h, ix.h, reads.h, p, counts.as_mut_ptr()));
        counts.truncate(n);
        Ok(counts)
    }

    /// estimate.rs:142-157 over a whole vector (f32, one rounding per operation, +inf for zero overlaps).
    pub fn estimates(&self, counts: &[u32], lens: &[u32], avg_target_len: f32, n_target: u64, thr: u32) -> crate::Result<Vec<f32>> {
        let mut out = vec![0f32; counts.len().max(1)];
        check!(self.h, lrge_hip_estimates(self.h, counts.as_ptr(), lens.as_ptr(), counts.len() as u32, avg_target_len,
                                          n_target, thr, out.as_mut_ptr()));
        out.truncate(counts.len());
        Ok(out)
    }
}

impl Ctx {
    /// Every chain of every read of `queries` against `ix` (two calls: the count, then the records) and the per-query seed
    /// statistics that complete the tags (rl; avg_k for dv).
    pub fn chains(&self, ix: &Index, queries: &SeqSet, dual: bool) -> crate::Result<(Vec<lrge_hip_chain>, Vec<i32>, Vec<u64>, Vec<u32>)> {
        let mut n: u64 = 0;
        check!(self.h, lrge_hip_chains(self.h, ix.h, queries.h, dual as c_int, ptr::null_mut(), 0, &mut n));
        let mut out = vec![lrge_hip_chain::default(); (n as usize).max(1)];
        check!(self.h, lrge_hip_chains(self.h, ix.h, queries.h, dual as c_int, out.as_mut_ptr(), n, &mut n));
        out.truncate(n as usize);
        let nq = queries.n as usize;
        let (mut rl, mut ss, mut nk) = (vec![0i32; nq.max(1)], vec![0u64; nq.max(1)], vec![0u32; nq.max(1)]);
        check!(self.h, lrge_hip_paf_stats(self.h, ix.h, queries.h, rl.as_mut_ptr(), ss.as_mut_ptr(), nk.as_mut_ptr()));
        Ok((out, rl, ss, nk))
    }

    /// `overlaps.paf`, where and as the reference writes it: `tmpdir/overlaps.paf` (twoset.rs:244-250, :410-416, ava.rs:220-226), one
    /// tab-separated line per mapping in `PafRecord`'s field order (mapping.rs:10-54) with its serialisers (mapping.rs:81-177:
    /// `tp:A:S`, `cm:i:`, `s1:i:`, `dv:f:` as `0` below f32::EPSILON else four decimals, `rl:i:`), EVERY mapping -- the `-F` filter
    /// only affects the counts (twoset.rs:293-301 serialises before it tests `is_internal`).  Line order is undefined in the
    /// reference (rayon workers under a mutex); here it is the library's chain order.  `q` / `t`: the streamed and the indexed set.
    pub fn write_overlaps_paf(&self, tmpdir: &Path, ix: &Index, qs: &SeqSet, q: &ReadSet, t: &ReadSet, dual: bool) -> crate::Result<()> {
        use std::io::Write;
        let (chains, rl, ss, nk) = self.chains(ix, qs, dual)?;
        let (q_lens, t_lens) = (q.lens(), t.lens());
        let file = std::fs::File::create(tmpdir.join("overlaps.paf"))?;
        let mut w = std::io::BufWriter::new(file);
        for c in &chains {
            let (qi, ti) = (c.query as usize, c.target as usize);
            let dv = chain_dv(c, q_lens[qi], t_lens[ti], ss[qi], nk[qi]);
            let dv_s = if dv < f32::EPSILON { "0".to_string() } else { format!("{dv:.4}") };
            writeln!(w, "{}\t{}\t{}\t{}\t{}\t{}\t{}\t{}\t{}\t{}\t{}\t0\ttp:A:S\tcm:i:{}\ts1:i:{}\tdv:f:{}\trl:i:{}",
                     String::from_utf8_lossy(&q.names[qi]), q_lens[qi], c.qs, c.qe, if c.rev != 0 { '-' } else { '+' },
                     String::from_utf8_lossy(&t.names[ti]), t_lens[ti], c.rs, c.re, c.mlen, c.blen, c.cnt, c.score, dv_s, rl[qi])
                .map_err(|e| LrgeError::PafWriteError(e.to_string()))?;
        }
        w.flush().map_err(|e| LrgeError::PafWriteError(e.to_string()))?;
        Ok(())
    }
}

/// mm_est_err (mm2:esterr.c) for one chain: n_match = cnt, n_tot = the kept seeds the chain spans plus the two end corrections,
/// avg_k = (float)sum_span / n_kept; dv = (float)(1.0 - pow((double)n_match / n_tot, 1.0 / avg_k)), 0 when n_match >= n_tot,
/// -1 when the query kept no seed (include/lrge_hip.h: lrge_hip_paf_stats).
fn chain_dv(c: &lrge_hip_chain, qlen: u32, tlen: u32, sum_span: u64, n_kept: u32) -> f32 {
    if n_kept == 0 { return -1.0; }
    let avg_k = sum_span as f32 / n_kept as f32;
    let n_match = c.cnt as i64;
    let mut n_tot = c.n_seeds as i64;
    if c.qs as f32 > avg_k && c.rs as f32 > avg_k { n_tot += 1; }
    if (qlen as i32 - c.qs) as f32 > avg_k && (tlen as i32 - c.re) as f32 > avg_k { n_tot += 1; }
    if n_match >= n_tot { return 0.0; }
    (1.0 - (n_match as f64 / n_tot as f64).powf(1.0 / avg_k as f64)) as f32
}

impl Drop for Ctx { fn drop(&mut self) { unsafe { lrge_hip_ctx_destroy(self.h) } } }
impl Drop for SeqSet<'_> { fn drop(&mut self) { unsafe { lrge_hip_seqset_free(self.h) } } }
impl Drop for Index<'_> { fn drop(&mut self) { unsafe { lrge_hip_index_free(self.h) } } }

// ------------------------------------------------------------------------------------------------------------------
// reads in memory
// ------------------------------------------------------------------------------------------------------------------
/// The (id, seq) records of one of the FASTA files `split_fastq` / `subsample_reads` wrote, concatenated.
pub struct ReadSet {
    pub bases: Vec<u8>,
    pub offsets: Vec<u64>, // n + 1
    pub names: Vec<Vec<u8>>,
}

impl ReadSet {
    pub fn len(&self) -> usize { self.names.len() }
    pub fn is_empty(&self) -> bool { self.names.is_empty() }
    pub fn lens(&self) -> Vec<u32> { self.offsets.windows(2).map(|w| (w[1] - w[0]) as u32).collect() }
}

/// io.rs:149-184 does the parsing (`read_id` = header up to the first whitespace, io.rs:199-204).
pub fn read_set<P: AsRef<Path>>(path: P) -> crate::Result<ReadSet> {
    let mut rs = ReadSet { bases: Vec::new(), offsets: vec![0], names: Vec::new() };
    iter_records(path, |id, seq| {
        rs.bases.extend_from_slice(seq);
        rs.offsets.push(rs.bases.len() as u64);
        rs.names.push(id.to_vec());
        Ok(())
    })?;
    Ok(rs)
}

/// Lexicographic (byte-wise = strcmp) ranks over the union of several name lists; equal names share a rank.
/// Replaces minimap2's `strcmp(qname, tname)` (skip_seed) and the name keys of liblrge's HashSet / HashMap.
pub fn name_ranks(lists: &[&[Vec<u8>]]) -> Vec<Vec<u32>> {
    let mut all: Vec<(&[u8], usize, usize)> = Vec::new();
    for (li, l) in lists.iter().enumerate() {
        for (i, n) in l.iter().enumerate() { all.push((n.as_slice(), li, i)); }
    }
    all.sort_by(|a, b| a.0.cmp(b.0));
    let mut out: Vec<Vec<u32>> = lists.iter().map(|l| vec![0u32; l.len()]).collect();
    let mut rank = 0u32;
    for j in 0..all.len() {
        if j > 0 && all[j].0 != all[j - 1].0 { rank = j as u32; }
        out[all[j].1][all[j].2] = rank;
    }
    out
}

// ------------------------------------------------------------------------------------------------------------------
// the two replacement bodies
// ------------------------------------------------------------------------------------------------------------------
pub struct TwoSetJob<'a> {
    pub target_file: &'a Path,
    pub query_file: &'a Path,
    pub avg_target_len: f32,      // split_fastq's third return value (twoset.rs:191)
    pub target_num_reads: usize,  // possibly reduced at twoset.rs:148
    pub query_num_reads: usize,
    pub target_num_bases: usize,
    pub query_num_bases: usize,
    pub remove_internal: bool,
    pub max_overhang_ratio: f32,
    pub use_min_ref: bool,
    pub pacbio: bool,             // Platform::PacBio -> Preset::AvaPb, else AvaOnt (twoset.rs:590-593)
    pub device: i32,
    pub tmpdir: &'a Path,         // self.tmpdir: `overlaps.paf` is written there (twoset.rs:244; kept with -C)
}

/// Body of `impl Estimate for TwoSetStrategy { fn generate_estimates }` after `split_fastq` (twoset.rs:587-606):
/// returns (per-read estimates, no_mapping_count).  The estimates are in query-file order (the reference returns them
/// in completion / HashMap order; only the multiset is defined, estimate.rs:63-76 sorts it anyway).
pub fn twoset_estimates(job: &TwoSetJob) -> crate::Result<(Vec<f32>, u32)> {
    let preset = if job.pacbio { LRGE_PRESET_AVA_PB } else { LRGE_PRESET_AVA_ONT };
    let t = read_set(job.target_file)?;
    let q = read_set(job.query_file)?;
    let ranks = name_ranks(&[&t.names, &q.names]);
    let (t_rank, q_rank) = (&ranks[0], &ranks[1]);
    let ctx = Ctx::new(job.device)?;
    let ts = ctx.upload(&t, t_rank)?;
    let qs = ctx.upload(&q, q_rank)?;
    let p = lrge_hip_params { remove_internal: job.remove_internal as i32, max_overhang_ratio: job.max_overhang_ratio };
    let inverse = job.use_min_ref && job.target_num_bases > job.query_num_bases; // twoset.rs:596
    let q_lens = q.lens();
    let (counts, no_mapping) = if inverse {
        ctx.presketch(&ts, preset)?;            // the streamed set is sketched beside the index build
        let ix = ctx.index(&qs, preset)?;       // index = query set (twoset.rs:597-599)
        let counts = ctx.overlap_inverse(&ix, q.len(), &ts, &p)?;
        ctx.write_overlaps_paf(job.tmpdir, &ix, &ts, &t, &q, true)?;    // twoset.rs:410-416,487: the streamed target read is the PAF's query
        let no_mapping = counts.iter().filter(|&&c| c == 0).count() as u32; // twoset.rs:545-569
        (counts, no_mapping)
    } else {
        ctx.presketch(&qs, preset)?;
        let ix = ctx.index(&ts, preset)?;       // twoset.rs:601-603
        let (counts, has) = ctx.overlap_twoset(&ix, &qs, &p)?;
        ctx.write_overlaps_paf(job.tmpdir, &ix, &qs, &q, &t, true)?;    // twoset.rs:244-250,293
        let no_mapping = has.iter().filter(|&&h| h == 0).count() as u32;    // twoset.rs:303-309
        (counts, no_mapping)
    };
    // per_read_estimate(read_len, avg_target_len, target_num_reads, n_overlaps, min_chain_score = 100)
    // (twoset.rs:319-331 forward, :557-564 inverse: the indexed QUERY read's length, the TARGET set's average and size)
    let est = ctx.estimates(&counts, &q_lens, job.avg_target_len, job.target_num_reads as u64, 100)?;
    Ok((est, no_mapping))
}

pub struct AvaJob<'a> {
    pub reads_file: &'a Path,
    pub num_reads: usize,         // after the clamp of ava.rs:122-128
    pub sum_len: usize,           // subsample_reads' second return value
    pub remove_internal: bool,
    pub max_overhang_ratio: f32,
    pub pacbio: bool,
    pub device: i32,
    pub tmpdir: &'a Path,         // `overlaps.paf` (ava.rs:220-226)
}

/// Body of `impl Estimate for AvaStrategy { fn generate_estimates }` after `subsample_reads` (ava.rs:369-381).
pub fn ava_estimates(job: &AvaJob) -> crate::Result<(Vec<f32>, u32)> {
    let preset = if job.pacbio { LRGE_PRESET_AVA_PB } else { LRGE_PRESET_AVA_ONT };
    let r = read_set(job.reads_file)?;
    // ava.rs:195-199: duplicate identifiers are an error; the library reports LRGE_ERR_DUPLICATE_ID for equal ranks,
    // the name for the message is found here
    let mut seen: HashMap<&[u8], ()> = HashMap::with_capacity(r.len());
    for n in &r.names {
        if seen.insert(n.as_slice(), ()).is_some() {
            return Err(LrgeError::DuplicateReadIdentifier(String::from_utf8_lossy(n).into_owned()));
        }
    }
    let ranks = name_ranks(&[&r.names]);
    let ctx = Ctx::new(job.device)?;
    let rs = ctx.upload(&r, &ranks[0])?;
    ctx.presketch(&rs, preset)?;
    let ix = ctx.index(&rs, preset)?;
    let p = lrge_hip_params { remove_internal: job.remove_internal as i32, max_overhang_ratio: job.max_overhang_ratio };
    let counts = ctx.overlap_ava(&ix, &rs, &p)?;
    ctx.write_overlaps_paf(job.tmpdir, &ix, &rs, &r, &r, false)?;      // ava.rs:220-226,273 (NO_DUAL: every pair once)
    let n_target = job.num_reads - 1;                                  // ava.rs:339-346
    let avg = job.sum_len as f32 / n_target as f32;                    // the read's own length is NOT subtracted
    let est = ctx.estimates(&counts, &r.lens(), avg, n_target as u64, 100)?;
    let no_mapping = counts.iter().filter(|&&c| c == 0).count() as u32; // ava.rs:329-331
    Ok((est, no_mapping))
}

// ------------------------------------------------------------------------------------------------------------------
// several GPUs from one process: one thread per GPU, the library's "local" communicator (lrge_hip.h)
// ------------------------------------------------------------------------------------------------------------------
/// Contiguous ranges of reads with (nearly) equal base counts: `world + 1` boundaries.
pub fn shard_by_bases(lens: &[u32], world: usize) -> Vec<usize> {
    let total: u64 = lens.iter().map(|&l| l as u64).sum();
    let mut bounds = vec![0usize];
    let (mut acc, mut i) = (0u64, 0usize);
    for r in 1..world {
        let goal = total * r as u64 / world as u64;
        while i < lens.len() && acc < goal { acc += lens[i] as u64; i += 1; }
        bounds.push(i);
    }
    bounds.push(lens.len());
    bounds
}

/// Two-set forward over `devices` with the TARGETS sharded (round 4; what pays when query bases x devices <= target bases, i.e. the
/// human-scale jobs): every rank uploads and indexes ITS contiguous share of the target reads and maps ALL queries against it;
/// `lrge_hip_index_build_tsharded` makes the occurrence statistics (mid_occ) those of the one index, the count vectors of the
/// ranks add up (disjoint targets: twoset.rs:286-317 counts distinct target names) in one all-reduce, and the estimates are
/// computed once from the summed counts.  No index entry crosses a link.
pub fn twoset_estimates_target_sharded(job: &TwoSetJob, devices: &[i32]) -> crate::Result<(Vec<f32>, u32)> {
    let world = devices.len();
    if world <= 1 { return twoset_estimates(job); }
    let preset = if job.pacbio { LRGE_PRESET_AVA_PB } else { LRGE_PRESET_AVA_ONT };
    let t = read_set(job.target_file)?;
    let q = read_set(job.query_file)?;
    let ranks = name_ranks(&[&t.names, &q.names]);
    let (q_lens, t_lens) = (q.lens(), t.lens());
    let t_bounds = shard_by_bases(&t_lens, world);
    if cross_shard_duplicates(&ranks[0], &t_bounds) { return twoset_estimates_multi(job, devices); }   // (names shared across shards: sharded by query instead)
    let mut group = ptr::null_mut();
    check!(ptr::null(), lrge_hip_comm_local_group_create(world as c_int, &mut group));
    let group = SendPtr(group);
    let p = lrge_hip_params { remove_internal: job.remove_internal as i32, max_overhang_ratio: job.max_overhang_ratio };
    let nq = q_lens.len();
    let results: Vec<crate::Result<(Vec<f32>, u32)>> = std::thread::scope(|sc| {
        let handles: Vec<_> = (0..world).map(|r| {
            let (t, q, ranks, q_lens, group, t_bounds) = (&t, &q, &ranks, &q_lens, &group, &t_bounds);
            let device = devices[r];
            sc.spawn(move || -> crate::Result<(Vec<f32>, u32)> {
                let ctx = Ctx::new(device)?;
                // from here on a `?` drops `comm` armed: the group is aborted and no peer waits for this rank
                let mut comm = Comm::local(&ctx, r, group)?;
                let (t0, t1) = (t_bounds[r], t_bounds[r + 1]);
                let tsub = ReadSet {
                    bases: t.bases[t.offsets[t0] as usize..t.offsets[t1] as usize].to_vec(),
                    offsets: t.offsets[t0..=t1].iter().map(|o| o - t.offsets[t0]).collect(),
                    names: t.names[t0..t1].to_vec(),
                };
                let ts = ctx.upload(&tsub, &ranks[0][t0..t1])?;
                let qs = ctx.upload(q, &ranks[1])?;                                 // ALL queries on every rank
                // ... sketched once per world: every rank runs mm_sketch over its share, the minimizers are all-gathered (collective)
                check!(ctx.h, lrge_hip_seqset_presketch_sharded(ctx.h, qs.h, preset, comm.h));
                let mut h = ptr::null_mut();
                check!(ctx.h, lrge_hip_index_build_tsharded(ctx.h, ts.h, preset, comm.h, &mut h));   // collective, and failure-collective
                let ix = Index { h, _ctx: &ctx };
                // The overlap call is this rank's own: if it fails the rank STILL enters the all-reduce that closes the step -- zeros
                // and a status word -- and returns its error afterwards; its peers see the word and fail too.  One vector
                // [counts | has_mapping | status]: disjoint targets, so the counts add up and has_mapping ORs.
                let mine = ctx.overlap_twoset(&ix, &qs, &p);
                let mut v = vec![0u32; 2 * nq + 1];
                match &mine {
                    Ok((counts, has)) => { v[..nq].copy_from_slice(counts); v[nq..2 * nq].copy_from_slice(has); }
                    Err(_) => v[2 * nq] = 1,
                }
                check!(ctx.h, lrge_hip_comm_allreduce_u32(comm.h, v.as_mut_ptr(), v.len()));
                comm.done();
                mine?;
                if v[2 * nq] != 0 { return Err(crate::LrgeError::ThreadError(format!("{} other GPU worker(s) failed in the overlap step", v[2 * nq]))); }
                let est = ctx.estimates(&v[..nq], q_lens, job.avg_target_len, job.target_num_reads as u64, 100)?;
                Ok((est, v[nq..2 * nq].iter().filter(|&&h| h == 0).count() as u32))
            })
        }).collect();
        handles.into_iter().map(|h| h.join().expect("GPU worker panicked")).collect()
    });
    unsafe { lrge_hip_comm_local_group_destroy(group.0) };
    let mut first = None;
    for r in results { let v = r?; if first.is_none() { first = Some(v); } }
    Ok(first.expect("world >= 2"))
}

/// A rank's communicator.  Dropped while still ARMED -- the rank is leaving through `?` between two collectives -- it aborts the
/// group first (`lrge_hip_comm_abort`): the peers blocked in, or heading for, the next collective return an error instead of waiting
/// for a rank that will never arrive, `thread::scope` joins, and the run ends with the failed rank's error, as a `MapError` ends the
/// reference's run (twoset.rs:279-284).
struct Comm { h: *mut lrge_hip_comm, armed: bool }
impl Comm {
    fn local(ctx: &Ctx, rank: usize, group: &SendPtr) -> crate::Result<Comm> {
        let mut h = ptr::null_mut();
        check!(ctx.h, lrge_hip_comm_create_local(ctx.h, rank as c_int, group.0, &mut h));
        Ok(Comm { h, armed: true })
    }
    /// the step's last collective has been left: nothing to abort any more
    fn done(&mut self) { self.armed = false; }
}
impl Drop for Comm {
    fn drop(&mut self) { unsafe { if self.armed { lrge_hip_comm_abort(self.h); } lrge_hip_comm_destroy(self.h) } }
}

/// True if a target identifier occurs in two DIFFERENT shards (`bounds` from `shard_by_bases`; equal names <=> equal rank).  The
/// shards' distinct-target counts add up only over disjoint names (twoset.rs:286-317 counts `target_name`s and never rejects a
/// duplicate id): such a set goes to `twoset_estimates_multi` (queries sharded) or to one GPU.
pub fn cross_shard_duplicates(target_ranks: &[u32], bounds: &[usize]) -> bool {
    let mut rp: Vec<(u32, u32)> = Vec::with_capacity(target_ranks.len());
    for s in 0..bounds.len() - 1 { for i in bounds[s]..bounds[s + 1] { rp.push((target_ranks[i], s as u32)); } }
    rp.sort_unstable();
    rp.windows(2).any(|w| w[0].0 == w[1].0 && w[0].1 != w[1].1)
}

struct SendPtr(*mut c_void);
unsafe impl Send for SendPtr {}
unsafe impl Sync for SendPtr {}

/// Two-set forward over `devices` (strong scaling of the one job, twoset.rs:266-334 sharded by query): every rank owns a
/// range of the queries end to end AND a contiguous share of the target reads, which is all it uploads, packs and sketches;
/// `lrge_hip_index_build_sharded` exchanges key sets, the entries each rank's queries can ask for and the hashes for the
/// global `mid_occ` (include/lrge_hip.h), one all-gather returns the estimates in query order.
pub fn twoset_estimates_multi(job: &TwoSetJob, devices: &[i32]) -> crate::Result<(Vec<f32>, u32)> {
    let world = devices.len();
    if world <= 1 { return twoset_estimates(job); }
    let preset = if job.pacbio { LRGE_PRESET_AVA_PB } else { LRGE_PRESET_AVA_ONT };
    let t = read_set(job.target_file)?;
    let q = read_set(job.query_file)?;
    let ranks = name_ranks(&[&t.names, &q.names]);
    let q_lens = q.lens();
    let t_lens = t.lens();
    let bounds = shard_by_bases(&q_lens, world);
    let t_bounds = shard_by_bases(&t_lens, world);
    let max_len = (0..world).map(|r| bounds[r + 1] - bounds[r]).max().unwrap_or(0);
    let mut group = ptr::null_mut();
    check!(ptr::null(), lrge_hip_comm_local_group_create(world as c_int, &mut group));
    let group = SendPtr(group);
    let p = lrge_hip_params { remove_internal: job.remove_internal as i32, max_overhang_ratio: job.max_overhang_ratio };
    let results: Vec<crate::Result<(Vec<f32>, u32)>> = std::thread::scope(|sc| {
        let handles: Vec<_> = (0..world).map(|r| {
            let (t, q, ranks, bounds, q_lens, group, t_lens, t_bounds) = (&t, &q, &ranks, &bounds, &q_lens, &group, &t_lens, &t_bounds);
            let device = devices[r];
            sc.spawn(move || -> crate::Result<(Vec<f32>, u32)> {
                let (lo, hi) = (bounds[r], bounds[r + 1]);
                let ctx = Ctx::new(device)?;
                let mut comm = Comm::local(&ctx, r, group)?;       // (armed: a `?` below aborts the group instead of leaving the peers waiting)
                let sub = ReadSet {
                    bases: q.bases[q.offsets[lo] as usize..q.offsets[hi] as usize].to_vec(),
                    offsets: q.offsets[lo..=hi].iter().map(|o| o - q.offsets[lo]).collect(),
                    names: q.names[lo..hi].to_vec(),
                };
                // this rank's share of the targets (shares in rank order: position lists keep the order of the one index)
                let (t0, t1) = (t_bounds[r], t_bounds[r + 1]);
                let tsub = ReadSet {
                    bases: t.bases[t.offsets[t0] as usize..t.offsets[t1] as usize].to_vec(),
                    offsets: t.offsets[t0..=t1].iter().map(|o| o - t.offsets[t0]).collect(),
                    names: t.names[t0..t1].to_vec(),
                };
                let ts = ctx.upload(&tsub, &ranks[0][t0..t1])?;
                let qs = ctx.upload(&sub, &ranks[1][lo..hi])?;
                let mut h = ptr::null_mut();
                check!(ctx.h, lrge_hip_index_build_sharded(ctx.h, t_lens.as_ptr(), ranks[0].as_ptr(), t_lens.len() as u32, ts.h,
                                                           t0 as u32, preset, qs.h, comm.h, &mut h));     // collective
                let ix = Index { h, _ctx: &ctx };
                let (counts, has) = ctx.overlap_twoset(&ix, &qs, &p)?;
                let est = ctx.estimates(&counts, &q_lens[lo..hi], job.avg_target_len, job.target_num_reads as u64, 100)?;
                let mut send = vec![f32::NAN; max_len.max(1)];
                send[..est.len()].copy_from_slice(&est);
                let mut recv = vec![0f32; max_len.max(1) * world];
                check!(ctx.h, lrge_hip_comm_allgather(comm.h, send.as_ptr() as *const c_void, send.len() * 4,
                                                      recv.as_mut_ptr() as *mut c_void));
                let mut nm = [has.iter().filter(|&&h| h == 0).count() as u32];
                check!(ctx.h, lrge_hip_comm_allreduce_u32(comm.h, nm.as_mut_ptr(), 1));
                comm.done();
                let m = max_len.max(1);
                let mut all = Vec::with_capacity(q_lens.len());
                for rr in 0..world { all.extend_from_slice(&recv[rr * m..rr * m + (bounds[rr + 1] - bounds[rr])]); }
                Ok((all, nm[0]))
            })
        }).collect();
        handles.into_iter().map(|h| h.join().expect("GPU worker panicked")).collect()
    });
    unsafe { lrge_hip_comm_local_group_destroy(group.0) };
    // every rank ends with the whole vector; rank 0's is returned (an error on any rank is the run's error)
    let mut first = None;
    for r in results { let v = r?; if first.is_none() { first = Some(v); } }
    Ok(first.expect("world >= 2"))
}
