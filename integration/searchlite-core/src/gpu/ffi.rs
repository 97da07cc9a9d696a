//! searchlite-core/src/gpu/ffi.rs — raw bindings of include/searchlite_gpu.h (libsearchlite_gpu.so).
//!
//! UNVERIFIED SOURCE: written against searchlite-core at the surveyed snapshot; the build image of
//! this repository has no cargo/rustc, so this file has never been compiled.  It is kept in sync
//! with the header by tests/test_abi.py (every declared function is bound here).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_float, c_int, c_void};

#[repr(C)] pub struct slg_index { _p: [u8; 0] }
#[repr(C)] pub struct slg_batch { _p: [u8; 0] }
#[repr(C)] pub struct slg_shard_group { _p: [u8; 0] }
#[repr(C)] pub struct slg_coalescer { _p: [u8; 0] }
/// slg_coalescer_submit / _wait: one request in flight (good for one wait).
#[repr(C)] #[derive(Clone, Copy)] pub struct slg_ticket { pub batch: *mut c_void, pub row: u32, pub k: u32, pub kind: u32 }

#[repr(C)]
pub struct slg_segment_desc {
    pub n_docs: u32, pub n_terms: u32,
    pub term_offsets: *const u64, pub doc_ids: *const u32, pub tfs: *const u32,
    pub term_field: *const u16,
    pub n_fields: u32, pub field_doc_len: *const *const c_float, pub field_avgdl: *const c_float,
    pub docs: c_float, pub k1: c_float, pub b: c_float,
    pub deleted: *const u8,
    pub vec_dim: u32, pub vec_metric: i32,
    pub vec_offsets: *const u32, pub vec_values: *const c_float, pub vec_rows: u32,
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct slg_tuning {
    pub struct_size: u32, pub validate: i32, pub champions: i32, pub allow_any_arch: i32, pub pruning: i32,
    pub uniform_max_terms: u32, pub uniform_round_target: u32, pub multi_round_target: u32,
    pub probe_target: u32, pub rounds_per_slice: u32, pub max_rounds_per_slice: u32,
    pub slices_per_subquery: u32, pub cand_mode: i32, pub slice_order: i32, pub block_max: i32,
    pub pool_cap_mb: u32, pub uniform_kernel: u32, pub uniform_sigma_x100: u32, pub inline_cuts: i32,
    pub updatable: i32, pub uniform_plans: i32, pub score_waves_per_simd: u32,
}
#[repr(C)] pub struct slg_vector_field_desc {
    pub vec_dim: u32, pub vec_metric: i32, pub vec_offsets: *const u32, pub vec_values: *const c_float, pub vec_rows: u32,
}
#[repr(C)] pub struct slg_score_plans {
    pub q_leaf: *const u32, pub q_plan: *const i32, pub q_tie: *const c_float, pub q_nleaves: *const u32,
    pub q_leaf_offsets: *const u32, pub leaf_group: *const u32, pub q_group_offsets: *const u32,
    pub group_plan: *const i32, pub group_tie: *const c_float,
    // trees of any shape: per query a node array in pre-order (SLG_PLAN_SUM | _DISMAX | _LEAF)
    pub q_node_offsets: *const u32, pub node_kind: *const i32, pub node_tie: *const c_float, pub node_parent: *const u32,
    pub q_min_match: *const u32,
}
#[repr(C)] pub struct slg_stats { pub scored_docs: u64, pub candidates_examined: u64, pub postings_advanced: u64 }
#[repr(C)] pub struct slg_query { pub n_terms: u32, pub term_ids: *const u32, pub weights: *const c_float }

#[link(name = "searchlite_gpu")]
extern "C" {
    pub fn slg_abi_version() -> u32;
    pub fn slg_last_error() -> *const c_char;
    pub fn slg_last_error_code() -> c_int;
    pub fn slg_tuning_default(out: *mut slg_tuning);
    pub fn slg_index_create_tuned(segs: *const slg_segment_desc, n_segs: u32, device: c_int,
        tuning_or_null: *const slg_tuning) -> *mut slg_index;
    pub fn slg_index_get_tuning(index: *const slg_index, out: *mut slg_tuning) -> c_int;
    pub fn slg_device_count() -> c_int;
    pub fn slg_index_create(segs: *const slg_segment_desc, n_segs: u32, device: c_int) -> *mut slg_index;
    pub fn slg_index_destroy(index: *mut slg_index);
    pub fn slg_index_info(index: *const slg_index, n_segs: *mut u32, n_postings: *mut u64, device_bytes: *mut u64) -> c_int;
    // index updates: the staged index follows the manifest (api/writer.rs:106-240)
    pub fn slg_index_update_deleted(index: *mut slg_index, seg: u32, deleted: *const u8, live_docs: c_float) -> c_int;
    pub fn slg_index_add_segment(index: *mut slg_index, seg: *const slg_segment_desc) -> c_int;
    pub fn slg_index_remove_segment(index: *mut slg_index, seg: u32) -> c_int;
    pub fn slg_index_generation(index: *const slg_index) -> u64;
    pub fn slg_index_device(index: *const slg_index) -> c_int;
    // request coalescer: concurrent single-query callers -> batches (searchlite-http/src/lib.rs:628-652)
    pub fn slg_coalescer_create(index: *mut slg_index, max_batch: u32, max_wait_us: u32) -> *mut slg_coalescer;
    pub fn slg_coalescer_destroy(coalescer: *mut slg_coalescer);
    pub fn slg_coalescer_search(coalescer: *mut slg_coalescer, query: *const slg_query, k: u32, strategy: c_int,
        out_doc: *mut u32, out_seg: *mut u32, out_score: *mut c_float, out_count: *mut u32,
        stats_or_null: *mut slg_stats) -> c_int;
    pub fn slg_coalescer_search_plan(coalescer: *mut slg_coalescer, query: *const slg_query, leaf: *const u32,
        plan: c_int, tie: c_float, n_leaves: u32, filter_id: i32, k: u32, strategy: c_int, out_doc: *mut u32,
        out_seg: *mut u32, out_score: *mut c_float, out_count: *mut u32, stats_or_null: *mut slg_stats) -> c_int;
    pub fn slg_coalescer_submit(coalescer: *mut slg_coalescer, query: *const slg_query, leaf: *const u32, plan: c_int,
                                tie: f32, n_leaves: u32, filter_id: i32, k: u32, strategy: c_int, want_stats: c_int,
                                ticket: *mut slg_ticket) -> c_int;
    pub fn slg_coalescer_poll(coalescer: *const slg_coalescer, ticket: *const slg_ticket) -> c_int;
    pub fn slg_coalescer_wait(coalescer: *mut slg_coalescer, ticket: *mut slg_ticket, out_doc: *mut u32,
                              out_seg: *mut u32, out_score: *mut f32, out_count: *mut u32,
                              stats_or_null: *mut slg_stats) -> c_int;
    pub fn slg_coalescer_last_error() -> *const c_char;
    pub fn slg_coalescer_phase_ms(coalescer: *const slg_coalescer, collect: *mut f64, prepare: *mut f64, run: *mut f64,
        fetch: *mut f64) -> c_int;
    pub fn slg_coalescer_stats(coalescer: *const slg_coalescer, n_batches: *mut u64, n_queries: *mut u64) -> c_int;
    // index sharding over RCCL (api/reader.rs:2670-2778 across GPUs)
    pub fn slg_shard_unique_id(out: *mut c_void, out_bytes: usize) -> c_int;
    pub fn slg_shard_group_create(index: *mut slg_index, rank: c_int, world: c_int, unique_id: *const c_void,
        segs_per_rank: u32) -> *mut slg_shard_group;
    pub fn slg_shard_group_destroy(group: *mut slg_shard_group);
    pub fn slg_batch_run_sharded(batch: *mut slg_batch, group: *mut slg_shard_group, out_doc: *mut u32,
        out_seg: *mut u32, out_score: *mut c_float, out_count: *mut u32) -> c_int;
    pub fn slg_batch_run_sharded_seq(batch: *mut slg_batch, group: *mut slg_shard_group, seq: u64, out_doc: *mut u32,
        out_seg: *mut u32, out_score: *mut c_float, out_count: *mut u32) -> c_int;
    pub fn slg_shard_group_skip_seq(group: *mut slg_shard_group, seq: u64) -> c_int;
    pub fn slg_shard_group_stats(group: *mut slg_shard_group, ms_kernels: *mut f64, ms_gather: *mut f64,
        ms_merge: *mut f64, n_runs: *mut u64) -> c_int;
    pub fn slg_batch_sharded_device_results(batch: *mut slg_batch, d_doc: *mut *mut c_void, d_seg: *mut *mut c_void,
        d_score: *mut *mut c_void, d_count: *mut *mut c_void) -> c_int;
    pub fn slg_batch_fetch_sharded(batch: *mut slg_batch, out_doc: *mut u32, out_seg: *mut u32,
        out_score: *mut c_float, out_count: *mut u32) -> c_int;
    pub fn slg_index_trim_pool(index: *mut slg_index, freed_bytes_or_null: *mut u64) -> c_int;
    pub fn slg_index_set_stream(index: *mut slg_index, hip_stream: *mut c_void) -> c_int;
    // doc filters: accept = !deleted && filter (api/reader.rs:3009-3018)
    pub fn slg_index_add_filter(index: *mut slg_index, seg_bitmaps: *const *const u8) -> c_int;
    pub fn slg_index_add_filter_terms(index: *mut slg_index, term_ids: *const u32, n_terms: u32, pass_if_absent: c_int,
                                      and_bitmaps_or_null: *const *const u8) -> c_int;
    pub fn slg_index_add_filter_range_i64(index: *mut slg_index, seg_columns: *const *const i64, lo: i64, hi: i64) -> c_int;
    pub fn slg_index_add_filter_range_f64(index: *mut slg_index, seg_columns: *const *const f64, lo: f64, hi: f64) -> c_int;
    pub fn slg_index_remove_filter(index: *mut slg_index, filter_id: c_int) -> c_int;
    // one-shot
    pub fn slg_search_batch(index: *mut slg_index, queries: *const slg_query, nq: u32, k: u32,
        strategy: c_int, out_doc: *mut u32, out_seg: *mut u32, out_score: *mut c_float,
        out_count: *mut u32, stats_or_null: *mut slg_stats) -> c_int;
    pub fn slg_search_batch_filtered(index: *mut slg_index, queries: *const slg_query, nq: u32,
        q_filter: *const i32, k: u32, strategy: c_int, out_doc: *mut u32, out_seg: *mut u32,
        out_score: *mut c_float, out_count: *mut u32, stats_or_null: *mut slg_stats) -> c_int;
    // prepared batches (CSR queries; optional score plan and filter per query)
    pub fn slg_batch_prepare(index: *mut slg_index, nq: u32, q_offsets: *const u32, q_term_ids: *const u32,
        q_weights: *const c_float, k: u32, strategy: c_int) -> *mut slg_batch;
    pub fn slg_batch_prepare_filtered(index: *mut slg_index, nq: u32, q_offsets: *const u32,
        q_term_ids: *const u32, q_weights: *const c_float, q_filter: *const i32, k: u32,
        strategy: c_int) -> *mut slg_batch;
    pub fn slg_batch_prepare_plan(index: *mut slg_index, nq: u32, q_offsets: *const u32,
        q_term_ids: *const u32, q_weights: *const c_float, q_leaf: *const u32, q_plan: *const i32,
        q_tie: *const c_float, q_nleaves: *const u32, q_filter: *const i32, k: u32,
        strategy: c_int) -> *mut slg_batch;
    pub fn slg_batch_prepare_plans(index: *mut slg_index, nq: u32, q_offsets: *const u32, q_term_ids: *const u32,
        q_weights: *const c_float, plans_or_null: *const slg_score_plans, q_filter_or_null: *const i32,
        k: u32, strategy: c_int) -> *mut slg_batch;
    pub fn slg_batch_set_stream(batch: *mut slg_batch, hip_stream: *mut c_void) -> c_int;
    pub fn slg_batch_run(batch: *mut slg_batch) -> c_int;
    pub fn slg_batch_sync(batch: *mut slg_batch) -> c_int;
    pub fn slg_batch_fetch(batch: *mut slg_batch, out_doc: *mut u32, out_seg: *mut u32, out_score: *mut c_float,
        out_count: *mut u32, stats_or_null: *mut slg_stats) -> c_int;
    pub fn slg_batch_device_results(batch: *mut slg_batch, d_doc: *mut *mut c_void, d_seg: *mut *mut c_void,
        d_score: *mut *mut c_void, d_count: *mut *mut c_void) -> c_int;
    pub fn slg_batch_device_result_block(batch: *mut slg_batch, d_block: *mut *mut c_void, n_bytes: *mut u64) -> c_int;
    pub fn slg_batch_info(batch: *const slg_batch, n_postings: *mut u64, n_slices: *mut u32, algorithmic_bytes: *mut u64) -> c_int;
    pub fn slg_batch_skip_counts(batch: *mut slg_batch, probed_postings: *mut u64, skipped_postings: *mut u64) -> c_int;
    pub fn slg_batch_destroy(batch: *mut slg_batch);
    // multi-GPU merge of per-shard result blocks, profiling, rerank
    pub fn slg_merge_shards_device(index: *mut slg_index, n_shards: u32, nq: u32, k: u32,
        d_doc: *const u32, d_seg: *const u32, d_score: *const c_float, d_count: *const u32, seg_stride: u32,
        d_out_doc: *mut u32, d_out_seg: *mut u32, d_out_score: *mut c_float, d_out_count: *mut u32) -> c_int;
    pub fn slg_profile_enable(index: *mut slg_index, on: c_int) -> c_int;
    pub fn slg_profile_read(index: *mut slg_index, n_launches: *mut u32, total_ms: *mut c_float) -> c_int;
    pub fn slg_rerank_batch(index: *mut slg_index, nq: u32, qvecs: *const c_float, alpha: *const c_float,
        cand_doc: *const u32, cand_seg: *const u32, cand_bm25: *const c_float, cand_count: *const u32,
        max_cand: u32, k_out: u32, out_doc: *mut u32, out_seg: *mut u32, out_score: *mut c_float,
        out_vec_score: *mut c_float, out_count: *mut u32) -> c_int;
    pub fn slg_rerank_batch_device(index: *mut slg_index, nq: u32, d_qvecs: *const c_float, d_alpha: *const c_float,
        d_cand_doc: *const u32, d_cand_seg: *const u32, d_cand_bm25: *const c_float, d_cand_count: *const u32,
        max_cand: u32, k_out: u32, d_out_doc: *mut u32, d_out_seg: *mut u32, d_out_score: *mut c_float,
        d_out_vec_score: *mut c_float, d_out_count: *mut u32) -> c_int;
    pub fn slg_batch_rerank_device(batch: *mut slg_batch, n_clauses: u32, d_qvecs: *const c_float, d_alpha: *const c_float,
        d_boost: *const c_float, k_out: u32, d_out_doc: *mut u32, d_out_seg: *mut u32, d_out_score: *mut c_float,
        d_out_vec_score: *mut c_float, d_out_count: *mut u32) -> c_int;
    pub fn slg_index_add_vector_field(index: *mut slg_index, per_segment: *const slg_vector_field_desc, n_segs: u32) -> c_int;
    pub fn slg_rerank_fields_batch(index: *mut slg_index, nq: u32, n_clauses: u32, clause_field: *const u32,
        qvecs: *const c_float, alpha: *const c_float, boost: *const c_float, cand_doc: *const u32, cand_seg: *const u32,
        cand_bm25: *const c_float, cand_count: *const u32, max_cand: u32, k_out: u32, out_doc: *mut u32,
        out_seg: *mut u32, out_score: *mut c_float, out_vec_score: *mut c_float, out_count: *mut u32) -> c_int;
    pub fn slg_rerank_fields_batch_device(index: *mut slg_index, nq: u32, n_clauses: u32, clause_field: *const u32,
        d_qvecs: *const c_float, d_alpha: *const c_float, d_boost: *const c_float, d_cand_doc: *const u32,
        d_cand_seg: *const u32, d_cand_bm25: *const c_float, d_cand_count: *const u32, max_cand: u32, k_out: u32,
        d_out_doc: *mut u32, d_out_seg: *mut u32, d_out_score: *mut c_float, d_out_vec_score: *mut c_float,
        d_out_count: *mut u32) -> c_int;
    pub fn slg_rerank_multi_batch(index: *mut slg_index, nq: u32, n_clauses: u32, qvecs: *const c_float,
        alpha: *const c_float, boost: *const c_float, cand_doc: *const u32, cand_seg: *const u32,
        cand_bm25: *const c_float, cand_count: *const u32, max_cand: u32, k_out: u32, out_doc: *mut u32,
        out_seg: *mut u32, out_score: *mut c_float, out_vec_score: *mut c_float, out_count: *mut u32) -> c_int;
    pub fn slg_rerank_multi_batch_device(index: *mut slg_index, nq: u32, n_clauses: u32, d_qvecs: *const c_float,
        d_alpha: *const c_float, d_boost: *const c_float, d_cand_doc: *const u32, d_cand_seg: *const u32,
        d_cand_bm25: *const c_float, d_cand_count: *const u32, max_cand: u32, k_out: u32, d_out_doc: *mut u32,
        d_out_seg: *mut u32, d_out_score: *mut c_float, d_out_vec_score: *mut c_float, d_out_count: *mut u32) -> c_int;
}
pub const SLG_OWN_STREAM: *mut c_void = usize::MAX as *mut c_void;
pub const SLG_NO_TERM: u32 = 0xFFFF_FFFF;
pub const SLG_METRIC_COSINE: i32 = 0;
pub const SLG_METRIC_L2: i32 = 1;
pub const SLG_STRATEGY_BM25: c_int = 0;
pub const SLG_STRATEGY_WAND: c_int = 1;
pub const SLG_STRATEGY_BMW: c_int = 2;
pub const SLG_PLAN_SUM: i32 = 0;
pub const SLG_PLAN_DISMAX: i32 = 1;
pub const SLG_PLAN_LEAF: i32 = 2;
pub const SLG_MAX_PLAN_DEPTH: usize = 4;
