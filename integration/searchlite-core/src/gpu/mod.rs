//! searchlite-core/src/gpu/mod.rs — the `gpu` cargo feature (searchlite-core/Cargo.toml:13,
//! src/lib.rs:11-12): MI355X batched BM25 top-k scorer + vector rerank behind IndexReader::search.
//!
//! UNVERIFIED SOURCE (no Rust toolchain in the build image of the GPU library; never compiled).
//! Written against these items of the surveyed snapshot:
//!   index/segment.rs:1328-1390   SegmentReader::{postings, avg_field_length, live_docs, is_deleted,
//!                                fast_fields, terms_with_prefix, vector_components}, `meta.doc_count`
//!   index/postings.rs:133-228    PostingsReader::{entries, len}, PostingEntry{doc_id, term_freq}
//!   index/fastfields.rs:683,1162 FastFieldsReader::i64_value, doc_length_key
//!   vectors/mod.rs:35-61         VectorStore::{dim, metric, offsets, values}
//!   api/reader.rs:904-910        QualifiedTerm{field, term, key, weight, leaf}
//!   api/reader.rs:2539-2906      IndexReader::search (call site, k = effective_limit + 1)
//!   api/reader.rs:2971-3000      term folding + ScoredTerm construction this module replaces
//!   api/reader.rs:3009-3036      accept = !deleted && matcher && filter && cursor
//!   query/planner.rs:88-165,207-213  QueryMatcher, QueryStringMatcher, ScoreExpr, ScorePlan, QueryPlan
//!   query/sort.rs:220-237        SortPlan::{is_score_only, primary_order}
//!   api/types.rs:396-439         SearchRequest
//!
//! Everything that is not eligible (below) keeps running on the CPU scorer; a non-zero return code
//! of the library also falls back to it — the GPU path never substitutes results silently.

pub mod ffi;
pub mod rerank;

use std::collections::HashMap;
use std::ffi::CStr;
use std::sync::Mutex;

use anyhow::{anyhow, bail, Result};

use crate::api::types::{ExecutionStrategy, Filter, SearchRequest, SortOrder};
use crate::index::fastfields::doc_length_key;
use crate::index::segment::SegmentReader;
use crate::query::filters::passes_filter;
use crate::query::planner::{QueryMatcher, QueryPlan, ScoreExpr};
use crate::query::sort::SortPlan;
use crate::DocId;

/// SLG_MAX_QUERY_TERMS / SLG_MAX_K of include/searchlite_gpu.h.
const MAX_QUERY_TERMS: usize = 32;
const MAX_K: usize = 20_001;

fn last_error() -> anyhow::Error {
  let msg = unsafe { CStr::from_ptr(ffi::slg_last_error()) }.to_string_lossy().into_owned();
  anyhow!("searchlite_gpu: {msg}")
}

/// All segments of an index staged in HBM (slg_index_create).  ONE per `Index`, not per reader:
/// searchlite opens a fresh IndexReader for every request (searchlite-http/src/lib.rs:640-643 ->
/// index/mod.rs:98-100 -> api/reader.rs:1887-1913), so the staged index lives on `InnerIndex`
/// (`gpu: Mutex<Option<Arc<GpuSegments>>>`) and `IndexReader::open` calls `GpuSegments::for_reader`,
/// which hands out the cached Arc while the manifest still describes what is staged and otherwise
/// brings the device index up to date IN PLACE (slg_index_update_deleted / _add_segment /
/// _remove_segment: a commit merges tombstones and appends at most one segment, api/writer.rs:150-192;
/// `docs = seg.live_docs()` feeds every idf, api/reader.rs:2985) — a full re-stage only when the
/// manifest changed in a way those three calls cannot express.
pub struct GpuSegments {
  handle: *mut ffi::slg_index,
  /// concurrent single-query callers -> batches (slg_coalescer_*)
  coalescer: *mut ffi::slg_coalescer,
  /// what is staged: (segment id, tombstones) per ordinal, and the library's generation after the
  /// last update.  A reader whose manifest snapshot differs falls back to the CPU scorer.
  state: std::sync::RwLock<StagedState>,
  /// filters already registered: serialized Filter -> filter id (slg_index_add_filter)
  filters: Mutex<FilterCache>,  // bounded LRU of registered doc filters
}

/// (segment id, number of tombstones) per segment ordinal — what `Manifest` says a reader sees.
pub(crate) type ManifestKey = Vec<(String, usize)>;

struct StagedState {
  key: ManifestKey,
  generation: u64,
  /// per segment: "field:term" key -> term id (rank of the key in the sorted dictionary)
  dict: Vec<std::sync::Arc<HashMap<String, u32>>>,
}

// The library serialises launches internally and planning takes no lock.
unsafe impl Send for GpuSegments {}
unsafe impl Sync for GpuSegments {}

impl Drop for GpuSegments {
  fn drop(&mut self) {
    unsafe {
      ffi::slg_coalescer_destroy(self.coalescer); // (no request can be inside: they hold an Arc)
      ffi::slg_index_destroy(self.handle)
    }
  }
}

pub(crate) fn manifest_key(segments: &[SegmentReader]) -> ManifestKey {
  segments.iter().map(|s| (s.meta.id.clone(), s.meta.deleted_docs.len())).collect()
}

fn tombstone_bitmap(seg: &SegmentReader) -> Vec<u8> {
  let n_docs = seg.meta.doc_count as usize;
  let mut bm = vec![0u8; n_docs.div_ceil(8)];
  for &d in seg.meta.deleted_docs.iter() {
    bm[(d >> 3) as usize] |= 1u8 << (d & 7);
  }
  bm
}

/// Host copies of one segment's arrays; they only have to outlive slg_index_create.
struct StagedSegment {
  term_offsets: Vec<u64>,
  doc_ids: Vec<u32>,
  tfs: Vec<u32>,
  term_field: Vec<u16>,
  field_lens: Vec<Option<Vec<f32>>>,
  field_len_ptrs: Vec<*const f32>,
  field_avgdl: Vec<f32>,
  deleted: Vec<u8>,
  vec_offsets: Vec<u32>,
  vec_values: std::sync::Arc<Vec<f32>>,
  vec_dim: u32,
  vec_metric: i32,
}

impl GpuSegments {
  /// What `IndexReader::open` calls with the cache slot of its `InnerIndex`: the staged index for
  /// this reader's manifest snapshot.  Cheap when nothing changed (a key compare + an Arc clone).
  pub fn for_reader(
    slot: &Mutex<Option<std::sync::Arc<GpuSegments>>>,
    segments: &[SegmentReader],
    fields: &[String],
    vector_field: Option<&str>,
    k1: f32,
    b: f32,
    device: i32,
  ) -> Option<std::sync::Arc<GpuSegments>> {
    let want = manifest_key(segments);
    let mut guard = slot.lock().unwrap();
    if let Some(g) = guard.as_ref() {
      if g.state.read().unwrap().key == want || g.follow(segments, &want, fields, vector_field, k1, b).is_ok() {
        return Some(g.clone());
      }
    }
    // first reader, or a manifest the update calls cannot reach from what is staged: stage afresh
    // (a missing GPU / library is not an error: the CPU scorer serves everything)
    let fresh = GpuSegments::stage(segments, fields, vector_field, k1, b, device).ok().map(std::sync::Arc::new);
    *guard = fresh.clone();
    fresh
  }

  /// Bring the device index from `state.key` to `want` with the update calls: tombstones that grew
  /// (same id at the same ordinal), segments appended at the end, segments that disappeared.
  fn follow(
    &self,
    segments: &[SegmentReader],
    want: &ManifestKey,
    fields: &[String],
    vector_field: Option<&str>,
    k1: f32,
    b: f32,
  ) -> Result<()> {
    let mut st = self.state.write().unwrap(); // no request builds term ids while the segment set changes
    // segments that left the manifest (compaction), highest ordinal first
    for ord in (0..st.key.len()).rev() {
      if !want.iter().any(|(id, _)| *id == st.key[ord].0) {
        if unsafe { ffi::slg_index_remove_segment(self.handle, ord as u32) } != 0 {
          return Err(last_error());
        }
        st.key.remove(ord);
        st.dict.remove(ord);
      }
    }
    // what is left must be a prefix of the manifest, in order
    if st.key.len() > want.len() || st.key.iter().zip(want.iter()).any(|(a, w)| a.0 != w.0 || a.1 > w.1) {
      bail!("manifest is not reachable from the staged index");
    }
    for ord in 0..st.key.len() {
      if st.key[ord].1 != want[ord].1 {
        let seg = &segments[ord];
        let bm = tombstone_bitmap(seg);
        if unsafe { ffi::slg_index_update_deleted(self.handle, ord as u32, bm.as_ptr(), seg.live_docs() as f32) } != 0 {
          return Err(last_error());
        }
        st.key[ord].1 = want[ord].1;
      }
    }
    for ord in st.key.len()..want.len() {
      let (staged, dict) = stage_one(&segments[ord], fields, vector_field)?;
      let desc = staged.descriptor(&segments[ord], fields.len() as u32, k1, b);
      if unsafe { ffi::slg_index_add_segment(self.handle, &desc) } < 0 {
        return Err(last_error());
      }
      st.key.push(want[ord].clone());
      st.dict.push(std::sync::Arc::new(dict));
    }
    st.generation = unsafe { ffi::slg_index_generation(self.handle) };
    // registered filters predate the new segments / tombstones of other segments: start over
    let mut cache = self.filters.lock().unwrap();
    for (_, (id, _)) in cache.map.drain() {
      unsafe { ffi::slg_index_remove_filter(self.handle, id) };
    }
    Ok(())
  }

  /// `fields`: every field that owns "field:term" keys (schema text fields, then keyword
  /// fields); `vector_field`: the field slg_rerank_* serves, if any.
  pub fn stage(
    segments: &[SegmentReader],
    fields: &[String],
    vector_field: Option<&str>,
    k1: f32,
    b: f32,
    device: i32,
  ) -> Result<Self> {
    let mut staged: Vec<StagedSegment> = Vec::with_capacity(segments.len());
    let mut dict: Vec<std::sync::Arc<HashMap<String, u32>>> = Vec::with_capacity(segments.len());
    for seg in segments {
      let (s, ids) = stage_one(seg, fields, vector_field)?;
      staged.push(s);
      dict.push(std::sync::Arc::new(ids));
    }
    let descs: Vec<ffi::slg_segment_desc> = segments
      .iter()
      .zip(staged.iter())
      .map(|(seg, s)| s.descriptor(seg, fields.len() as u32, k1, b))
      .collect();
    let handle = unsafe { ffi::slg_index_create(descs.as_ptr(), descs.len() as u32, device) };
    if handle.is_null() {
      return Err(last_error());
    }
    // batches of up to 1024 requests; a batch closes 30 us after its first request unless the device is idle
    let coalescer = unsafe { ffi::slg_coalescer_create(handle, 1024, 30) };
    if coalescer.is_null() {
      unsafe { ffi::slg_index_destroy(handle) };
      bail!("slg_coalescer_create failed");
    }
    let state = StagedState { key: manifest_key(segments), generation: 0, dict };
    Ok(Self { handle, coalescer, state: std::sync::RwLock::new(state), filters: Mutex::new(FilterCache::default()) })
  }
}

impl StagedSegment {
  fn descriptor(&self, seg: &SegmentReader, n_fields: u32, k1: f32, b: f32) -> ffi::slg_segment_desc {
    let s = self;
    ffi::slg_segment_desc {
      n_docs: seg.meta.doc_count,
      n_terms: s.term_field.len() as u32,
      term_offsets: s.term_offsets.as_ptr(),
      doc_ids: s.doc_ids.as_ptr(),
      tfs: s.tfs.as_ptr(),
      term_field: s.term_field.as_ptr(),
      n_fields,
      field_doc_len: s.field_len_ptrs.as_ptr(),
      field_avgdl: s.field_avgdl.as_ptr(),
      docs: seg.live_docs() as f32, // api/reader.rs:2985
      k1,
      b,
      deleted: s.deleted.as_ptr(),
      vec_dim: s.vec_dim,
      vec_metric: s.vec_metric,
      vec_offsets: if s.vec_dim > 0 { s.vec_offsets.as_ptr() } else { std::ptr::null() },
      vec_values: if s.vec_dim > 0 { s.vec_values.as_ptr() } else { std::ptr::null() },
      vec_rows: if s.vec_dim > 0 { (s.vec_values.len() / s.vec_dim as usize) as u32 } else { 0 },
    }
  }
}

/// Host arrays of ONE segment in the layout slg_segment_desc borrows, and its term dictionary.
fn stage_one(
  seg: &SegmentReader,
  fields: &[String],
  vector_field: Option<&str>,
) -> Result<(StagedSegment, HashMap<String, u32>)> {
  let field_id: HashMap<&str, u16> =
    fields.iter().enumerate().map(|(i, f)| (f.as_str(), i as u16)).collect();
  let _ = vector_field; // (read only with the `vectors` feature)
  let n_docs = seg.meta.doc_count as usize;
  let mut s = StagedSegment {
    term_offsets: vec![0],
    doc_ids: Vec::new(),
    tfs: Vec::new(),
    term_field: Vec::new(),
    field_lens: Vec::new(),
    field_len_ptrs: Vec::new(),
    field_avgdl: Vec::new(),
    deleted: vec![0u8; n_docs.div_ceil(8)],
    vec_offsets: Vec::new(),
    vec_values: std::sync::Arc::new(Vec::new()),
    vec_dim: 0,
    vec_metric: ffi::SLG_METRIC_COSINE,
  };
  let mut ids = HashMap::new();
  // the dictionary iterates in sorted key order (TinyFst is a BTreeMap, util/fst.rs:4-31)
  for key in seg.terms_with_prefix("") {
    let Some((field, _)) = key.split_once(':') else { continue };
    let Some(&fid) = field_id.get(field) else { continue };
    let Some(postings) = seg.postings(key) else { continue };
    ids.insert(key.clone(), s.term_field.len() as u32);
    s.term_field.push(fid);
    for e in postings.entries() {
      s.doc_ids.push(e.doc_id);
      s.tfs.push(e.term_freq);
    }
    s.term_offsets.push(s.doc_ids.len() as u64);
  }
  for f in fields {
    // field_lengths_for, api/reader.rs:3604-3621: absent -> 0 -> the scorer's max(avgdl, 1)
    let key = doc_length_key(f);
    let lens: Vec<f32> = (0..n_docs as DocId)
      .map(|d| seg.fast_fields().i64_value(&key, d).unwrap_or(0) as f32)
      .collect();
    s.field_avgdl.push(seg.avg_field_length(f));
    s.field_lens.push(Some(lens));
  }
  for d in 0..n_docs as DocId {
    if seg.is_deleted(d) {
      s.deleted[(d >> 3) as usize] |= 1u8 << (d & 7);
    }
  }
  #[cfg(feature = "vectors")]
  if let Some((_, store)) = vector_field.and_then(|vf| seg.vector_components(vf)) {
    s.vec_dim = store.dim() as u32;
    s.vec_metric = match store.metric() {
      crate::vectors::VectorMetric::Cosine => ffi::SLG_METRIC_COSINE,
      crate::vectors::VectorMetric::L2 => ffi::SLG_METRIC_L2,
    };
    s.vec_offsets = store.offsets().to_vec();
    s.vec_values = store.values();
  }
  s.field_len_ptrs =
    s.field_lens.iter().map(|l| l.as_ref().map_or(std::ptr::null(), |v| v.as_ptr())).collect();
  Ok((s, ids))
}

impl GpuSegments {
  pub(crate) fn raw(&self) -> *mut ffi::slg_index {
    self.handle
  }

  /// `req.filter` as a doc bitmap per segment (accept = !deleted && filter, api/reader.rs:
  /// 3009-3018), evaluated once with the reference's own passes_filter and cached by the filter's
  /// serialized form.
  ///
  /// `not_keys`: the term keys ("field:term", as the term dictionaries hold them) of the matcher's not-term
  /// groups (api/reader.rs:1499-1503: a doc that holds one of them never matches).  Their posting lists are
  /// already on the device: the filter is built there (slg_index_add_filter_terms), AND-ed with `filter`.
  fn filter_id(
    &self,
    segments: &[SegmentReader],
    filter: Option<&Filter>,
    not_keys: &[String],
    dict: &[std::sync::Arc<HashMap<String, u32>>],
  ) -> Result<i32> {
    let mut key = match filter {
      Some(f) => serde_json::to_string(f)?,
      None => String::new(),
    };
    if !not_keys.is_empty() {
      let mut sorted: Vec<&str> = not_keys.iter().map(|k| k.as_str()).collect();
      sorted.sort_unstable();
      sorted.dedup();
      key.push('\u{1}');
      key.push_str(&sorted.join("\u{1}"));
    }
    // the lock is held across lookup, evaluation and insert: two threads that miss on the same
    // filter must not both register it (each registration is n_docs / 8 bytes of HBM per segment)
    let mut cache = self.filters.lock().unwrap();
    cache.tick += 1;
    let now = cache.tick;
    if let Some(e) = cache.map.get_mut(&key) {
      e.1 = now;
      return Ok(e.0);
    }
    // bounded: high-cardinality filters (per-user ranges) would otherwise grow the reject table
    // and device memory without limit.  The least recently used filter leaves the device.
    if cache.map.len() >= MAX_CACHED_FILTERS {
      if let Some(old_key) = cache.map.iter().min_by_key(|(_, e)| e.1).map(|(k, _)| k.clone()) {
        if let Some((old_id, _)) = cache.map.remove(&old_key) {
          unsafe { ffi::slg_index_remove_filter(self.handle, old_id) };
        }
      }
    }
    let bitmaps: Vec<Vec<u8>> = match filter {
      Some(filter) => segments
        .iter()
        .map(|seg| {
          let n = seg.meta.doc_count as usize;
          let mut bm = vec![0u8; n.div_ceil(8)];
          for d in 0..n as DocId {
            if passes_filter(seg.fast_fields(), d, filter) {
              bm[(d >> 3) as usize] |= 1u8 << (d & 7);
            }
          }
          bm
        })
        .collect(),
      None => Vec::new(),
    };
    let ptrs: Vec<*const u8> = bitmaps.iter().map(|b| b.as_ptr()).collect();
    let id = if not_keys.is_empty() {
      unsafe { ffi::slg_index_add_filter(self.handle, ptrs.as_ptr()) }
    } else {
      // one row of per-segment term ids per not-term key (`dict`: the staged state's term dictionaries, which
      // the caller reads under the state's lock)
      let mut ids = Vec::with_capacity(not_keys.len() * dict.len());
      for k in not_keys {
        for d in dict.iter() {
          ids.push(d.get(k).copied().unwrap_or(ffi::SLG_NO_TERM));
        }
      }
      unsafe {
        ffi::slg_index_add_filter_terms(
          self.handle,
          ids.as_ptr(),
          not_keys.len() as u32,
          1,
          if ptrs.is_empty() { std::ptr::null() } else { ptrs.as_ptr() },
        )
      }
    };
    if id < 0 {
      return Err(last_error());
    }
    cache.map.insert(key, (id, now));
    Ok(id)
  }
}

/// Filters kept on the device per index (LRU beyond this).
const MAX_CACHED_FILTERS: usize = 256;

/// filter (serialized) -> (device filter id, last use)
#[derive(Default)]
pub(crate) struct FilterCache {
  map: HashMap<String, (i32, u64)>,
  tick: u64,
}

/// One scored term of a request, folded as search_segment does (api/reader.rs:2971-2983):
/// identical keys collapse, weights add, the first occurrence fixes the leaf.
pub(crate) struct FoldedTerm {
  pub key: String,
  pub weight: f32,
  pub leaf: u32,
}

pub(crate) fn fold_terms<'a>(qualified: impl Iterator<Item = (&'a str, f32, usize)>) -> Vec<FoldedTerm> {
  let mut order: Vec<FoldedTerm> = Vec::new();
  let mut pos: HashMap<&'a str, usize> = HashMap::new();
  for (key, weight, leaf) in qualified {
    match pos.get(key) {
      Some(&i) => order[i].weight += weight,
      None => {
        pos.insert(key, order.len());
        order.push(FoldedTerm { key: key.to_string(), weight, leaf: leaf as u32 });
      }
    }
  }
  order
}

/// The shape of ScorePlan the device evaluates (query/planner.rs:113-153): a root over leaves, or
/// (two levels) a root over groups of consecutive leaves.
pub(crate) enum GpuScorePlan {
  /// no plan, `Leaf`, or `Sum` of leaves: per-leaf sums added in leaf order
  Sum,
  /// `DisMax { children: leaves, tie_breaker }`
  DisMax { tie_breaker: f32 },
  /// root `Sum` | `DisMax` whose children are leaves or `Sum` / `DisMax` of leaves: what
  /// `dis_max{queries}` (planner.rs:470-487) and `bool{should: [multi_match ...]}` (:670-690) build.
  /// leaf_group[l] = group of leaf l; a bare leaf child is a Sum group of one leaf.
  Tree { root_dismax: bool, root_tie: f32, leaf_group: Vec<u32>, group_plan: Vec<i32>, group_tie: Vec<f32> },
  /// anything deeper (up to SLG_MAX_PLAN_DEPTH levels of Sum / DisMax above the leaves): the tree node
  /// by node in pre-order (slg_score_plans::q_node_offsets)
  Nodes { kind: Vec<i32>, tie: Vec<f32>, parent: Vec<u32> },
}

/// Pre-order node arrays of `e`; leaves must come out numbered 0, 1, 2, ... (the planner assigns leaf ids
/// in traversal order) and every Sum / DisMax needs a child.  `depth` = levels of internal nodes above `e`.
fn push_nodes(e: &ScoreExpr, parent: u32, depth: usize, next_leaf: &mut usize, kind: &mut Vec<i32>, tie: &mut Vec<f32>, par: &mut Vec<u32>) -> bool {
  let me = kind.len() as u32;
  match e {
    ScoreExpr::Leaf(i) => {
      if *i != *next_leaf {
        return false;
      }
      *next_leaf += 1;
      kind.push(ffi::SLG_PLAN_LEAF);
      tie.push(0.0);
      par.push(parent);
      true
    }
    ScoreExpr::Sum(cs) | ScoreExpr::DisMax { children: cs, .. } => {
      if cs.is_empty() || depth >= ffi::SLG_MAX_PLAN_DEPTH {
        return false;
      }
      let (k, t) = match e {
        ScoreExpr::DisMax { tie_breaker, .. } => (ffi::SLG_PLAN_DISMAX, *tie_breaker),
        _ => (ffi::SLG_PLAN_SUM, 0.0),
      };
      kind.push(k);
      tie.push(t);
      par.push(parent);
      cs.iter().all(|c| push_nodes(c, me, depth + 1, next_leaf, kind, tie, par))
    }
  }
}

fn plan_shape(plan: &QueryPlan) -> Option<(GpuScorePlan, u32)> {
  let Some(sp) = plan.scorer.as_ref() else { return Some((GpuScorePlan::Sum, 0)) };
  let all_leaves = |cs: &[ScoreExpr]| cs.iter().all(|c| matches!(c, ScoreExpr::Leaf(_)));
  let n_leaves = sp.leaf_count as u32;
  let (children, root_dismax, root_tie) = match &sp.root {
    ScoreExpr::Leaf(_) => return Some((GpuScorePlan::Sum, n_leaves)),
    ScoreExpr::Sum(cs) if all_leaves(cs) => return Some((GpuScorePlan::Sum, n_leaves)),
    ScoreExpr::DisMax { children, tie_breaker } if all_leaves(children) && !children.is_empty() => {
      return Some((GpuScorePlan::DisMax { tie_breaker: *tie_breaker }, n_leaves));
    }
    ScoreExpr::Sum(cs) => (cs.as_slice(), false, 0.0f32),
    ScoreExpr::DisMax { children, tie_breaker } => (children.as_slice(), true, *tie_breaker),
  };
  // two levels: every child a leaf or a Sum / DisMax of leaves, the leaves numbered in traversal
  // order (so a group's leaves are consecutive and every leaf belongs to exactly one group)
  let mut leaf_group = vec![u32::MAX; sp.leaf_count];
  let (mut group_plan, mut group_tie) = (Vec::new(), Vec::new());
  let mut next_leaf = 0usize;
  for (g, child) in children.iter().enumerate() {
    let (leaves, kind, tie): (Vec<usize>, i32, f32) = match child {
      ScoreExpr::Leaf(i) => (vec![*i], ffi::SLG_PLAN_SUM, 0.0),
      ScoreExpr::Sum(cs) if all_leaves(cs) && !cs.is_empty() => {
        (cs.iter().map(|c| if let ScoreExpr::Leaf(i) = c { *i } else { unreachable!() }).collect(), ffi::SLG_PLAN_SUM, 0.0)
      }
      ScoreExpr::DisMax { children: cs, tie_breaker } if all_leaves(cs) && !cs.is_empty() => (
        cs.iter().map(|c| if let ScoreExpr::Leaf(i) = c { *i } else { unreachable!() }).collect(),
        ffi::SLG_PLAN_DISMAX,
        *tie_breaker,
      ),
      _ => {
        // three levels or more: the whole tree node by node
        let (mut kind, mut tie, mut par) = (Vec::new(), Vec::new(), Vec::new());
        let mut next = 0usize;
        if push_nodes(&sp.root, 0, 0, &mut next, &mut kind, &mut tie, &mut par) && next == sp.leaf_count {
          return Some((GpuScorePlan::Nodes { kind, tie, parent: par }, n_leaves));
        }
        return None;
      }
    };
    for l in leaves {
      if l != next_leaf || l >= leaf_group.len() {
        return None;
      }
      leaf_group[l] = g as u32;
      next_leaf += 1;
    }
    group_plan.push(kind);
    group_tie.push(tie);
  }
  if next_leaf != sp.leaf_count || group_plan.len() > MAX_QUERY_TERMS {
    return None;
  }
  Some((GpuScorePlan::Tree { root_dismax, root_tie, leaf_group, group_plan, group_tie }, n_leaves))
}

/// "the doc has a posting of some scored term" implies the matcher (QueryEvaluator::matches,
/// api/reader.rs:1486-1518): a term, a query string of term groups, or a dis_max / bool.should of such
fn pure_disjunction(m: &QueryMatcher) -> bool {
  match m {
    QueryMatcher::Term(_) => true,
    QueryMatcher::QueryString(qs) => {
      // (minimum_should_match > 1 only at the top of the matcher: see min_should_match below)
      !qs.term_groups.is_empty()
        && qs.phrase_groups.is_empty()
        && qs.not_term_groups.is_empty()
        && qs.minimum_should_match.unwrap_or(1) <= 1
    }
    QueryMatcher::DisMax(children) => !children.is_empty() && children.iter().all(pure_disjunction),
    QueryMatcher::Bool { must, should, must_not, filter, minimum_should_match } => {
      must.is_empty()
        && must_not.is_empty()
        && filter.is_empty()
        && !should.is_empty()
        && minimum_should_match.unwrap_or(1) <= 1
        && should.iter().all(pure_disjunction)
    }
    _ => false,
  }
}

/// minimum_should_match of a request whose whole matcher is ONE query string of term groups
/// (api/reader.rs:1509-1517: `matched_terms >= required`): the device counts, per doc, the ScorePlan leaves
/// that hold it (slg_score_plans::q_min_match) — a term group is a leaf (query/planner.rs:354-360).
/// Some(1) for every other pure disjunction; None: a shape the device does not count (CPU scorer).
fn min_should_match(m: &QueryMatcher) -> Option<u32> {
  match m {
    // (not-term groups reach the device as a filter built from their posting lists: not_term_keys below)
    QueryMatcher::QueryString(qs) if !qs.term_groups.is_empty() && qs.phrase_groups.is_empty() => {
      let need = qs.minimum_should_match.unwrap_or(1);
      if need <= 255 { Some(need.max(1) as u32) } else { None }
    }
    other => if pure_disjunction(other) { Some(1) } else { None },
  }
}

/// is this request one the GPU scorer reproduces exactly?
/// `needs_score_hook` = has_custom_scoring(&compiled_score) (api/reader.rs:376-387, :2628).
pub(crate) fn gpu_eligible(
  req: &SearchRequest,
  sort_plan: &SortPlan,
  plan: &QueryPlan,
  needs_score_hook: bool,
  top_k: usize,
  n_folded_terms: usize,
) -> Option<(GpuScorePlan, u32, u32)> {
  // score_fast_path (api/reader.rs:2550-2551): sort = _score desc only => ScoreMode::Score and
  // the scorer is handed rank_limit = top_k (:2702-2703)
  let score_fast_path =
    sort_plan.is_score_only() && matches!(sort_plan.primary_order(), Some(SortOrder::Desc));
  if !score_fast_path || !req.return_hits || req.limit == 0 || top_k == 0 || top_k > MAX_K {
    return None;
  }
  // no collector: agg_ref stays None only without aggregations (api/reader.rs:2694-2699)
  if !req.aggs.is_empty() || req.explain || needs_score_hook || req.cursor.is_some() {
    return None;
  }
  if req.collapse.is_some() {
    return None; // collapse needs every hit of a group, not just the top k
  }
  if n_folded_terms == 0 || n_folded_terms > MAX_QUERY_TERMS {
    return None;
  }
  let min_match = min_should_match(&plan.matcher)?;
  if !plan.phrase_specs.is_empty() {
    return None;
  }
  // every matching term group must also score, else a doc could match without a scored posting
  // (the not-term groups of a top-level query string do not score and do not match: they only reject)
  let not_groups: &[usize] = match &plan.matcher {
    QueryMatcher::QueryString(qs) => &qs.not_term_groups,
    _ => &[],
  };
  if plan.term_groups.iter().enumerate().any(|(i, g)| !g.score && !not_groups.contains(&i)) {
    return None;
  }
  let (shape, n_leaves) = plan_shape(plan)?;
  // the device counts leaves in the few-term kernel's plan instantiation only: a flat plan, <= 8 scored lists
  if min_match > 1
    && !(matches!(shape, GpuScorePlan::Sum | GpuScorePlan::DisMax { .. }) && n_folded_terms <= 8)
  {
    return None;
  }
  Some((shape, n_leaves, min_match))
}

/// Replaces the per-segment loop + cross-segment sort of IndexReader::search
/// (api/reader.rs:2670-2778) for ONE eligible request: (segment_ord, doc_id, score) in final order,
/// at most `top_k` of them, plus the number of distinct docs scored.
pub(crate) fn gpu_top_k(
  gpu: &GpuSegments,
  segments: &[SegmentReader],
  folded: &[FoldedTerm],
  score_plan: &GpuScorePlan,
  n_leaves: u32,
  min_match: u32,
  filter: Option<&Filter>,
  not_keys: &[String],
  execution: &ExecutionStrategy,
  top_k: usize,
) -> Result<(Vec<(u32, DocId, f32)>, u64)> {
  let n_segs = segments.len();
  // the reader's manifest snapshot must be what is staged (a commit may have moved the device index
  // on since this reader opened: such a request runs on the CPU scorer, as does the next one only if
  // its reader is older still).  The read lock keeps the segment set fixed while the ids are built
  // and the query is handed over.
  let st = gpu.state.read().unwrap();
  if st.key != manifest_key(segments) {
    bail!("staged index has moved past this reader's manifest");
  }
  let mut term_ids = Vec::with_capacity(folded.len() * n_segs);
  for t in folded {
    for d in st.dict.iter() {
      term_ids.push(d.get(&t.key).copied().unwrap_or(ffi::SLG_NO_TERM));
    }
  }
  let weights: Vec<f32> = folded.iter().map(|t| t.weight).collect();
  let leaves: Vec<u32> = folded.iter().map(|t| t.leaf).collect();
  let offsets = [0u32, folded.len() as u32];
  let (plan_kind, tie) = match score_plan {
    GpuScorePlan::Sum => (ffi::SLG_PLAN_SUM, 0.0f32),
    GpuScorePlan::DisMax { tie_breaker } => (ffi::SLG_PLAN_DISMAX, *tie_breaker),
    GpuScorePlan::Tree { root_dismax, root_tie, .. } => {
      (if *root_dismax { ffi::SLG_PLAN_DISMAX } else { ffi::SLG_PLAN_SUM }, *root_tie)
    }
    GpuScorePlan::Nodes { .. } => (ffi::SLG_PLAN_SUM, 0.0f32), // (not read: the node arrays carry the root)
  };
  let node_offsets = match score_plan {
    GpuScorePlan::Nodes { kind, .. } => [0u32, kind.len() as u32],
    _ => [0u32, 0],
  };
  // two-level plans: CSR of one query (slg_score_plans)
  let (leaf_offsets, group_offsets) = match score_plan {
    GpuScorePlan::Tree { leaf_group, group_plan, .. } => ([0u32, leaf_group.len() as u32], [0u32, group_plan.len() as u32]),
    _ => ([0u32, 0], [0u32, 0]),
  };
  let plans = ffi::slg_score_plans {
    q_leaf: leaves.as_ptr(),
    q_plan: &plan_kind,
    q_tie: &tie,
    q_nleaves: if n_leaves > 0 { &n_leaves } else { std::ptr::null() },
    q_leaf_offsets: leaf_offsets.as_ptr(),
    leaf_group: match score_plan {
      GpuScorePlan::Tree { leaf_group, .. } => leaf_group.as_ptr(),
      _ => std::ptr::null(),
    },
    q_group_offsets: group_offsets.as_ptr(),
    group_plan: match score_plan {
      GpuScorePlan::Tree { group_plan, .. } => group_plan.as_ptr(),
      _ => std::ptr::null(),
    },
    group_tie: match score_plan {
      GpuScorePlan::Tree { group_tie, .. } => group_tie.as_ptr(),
      _ => std::ptr::null(),
    },
    q_node_offsets: match score_plan {
      GpuScorePlan::Nodes { .. } => node_offsets.as_ptr(),
      _ => std::ptr::null(),
    },
    node_kind: match score_plan {
      GpuScorePlan::Nodes { kind, .. } => kind.as_ptr(),
      _ => std::ptr::null(),
    },
    node_tie: match score_plan {
      GpuScorePlan::Nodes { tie, .. } => tie.as_ptr(),
      _ => std::ptr::null(),
    },
    node_parent: match score_plan {
      GpuScorePlan::Nodes { parent, .. } => parent.as_ptr(),
      _ => std::ptr::null(),
    },
    q_min_match: if min_match > 1 { &min_match } else { std::ptr::null() },
  };
  let filter_id = if filter.is_some() || !not_keys.is_empty() {
    gpu.filter_id(segments, filter, not_keys, &st.dict)?
  } else {
    -1
  };
  let strategy = match execution {
    ExecutionStrategy::Bm25 => ffi::SLG_STRATEGY_BM25,
    ExecutionStrategy::Wand => ffi::SLG_STRATEGY_WAND,
    ExecutionStrategy::Bmw => ffi::SLG_STRATEGY_BMW,
  };
  let k = top_k as u32;
  // Flat plans (every query string, multi_match, dis_max of terms) go through the coalescer: the
  // reference has no batch API (api/reader.rs:2539) and serves a request per blocking thread
  // (searchlite-http/src/lib.rs:628-652), so concurrent requests share one prepare / run / fetch.
  // (a request with minimum_should_match > 1 is prepared on its own: the coalescer's rows carry no such count)
  if min_match <= 1 && matches!(score_plan, GpuScorePlan::Sum | GpuScorePlan::DisMax { .. }) {
    let (mut doc, mut seg, mut score) = (vec![0u32; top_k], vec![0u32; top_k], vec![0f32; top_k]);
    let mut count = 0u32;
    let mut stats = ffi::slg_stats { scored_docs: 0, candidates_examined: 0, postings_advanced: 0 };
    let q = ffi::slg_query { n_terms: folded.len() as u32, term_ids: term_ids.as_ptr(), weights: weights.as_ptr() };
    let rc = unsafe {
      ffi::slg_coalescer_search_plan(
        gpu.coalescer, &q, leaves.as_ptr(), plan_kind, tie, n_leaves, filter_id, k, strategy,
        doc.as_mut_ptr(), seg.as_mut_ptr(), score.as_mut_ptr(), &mut count, &mut stats,
      )
    };
    drop(st);
    if rc != 0 {
      bail!("searchlite_gpu returned {rc}");
    }
    let hits = (0..count as usize).map(|i| (seg[i], doc[i] as DocId, score[i])).collect();
    return Ok((hits, stats.scored_docs));
  }
  let batch = unsafe {
    ffi::slg_batch_prepare_plans(
      gpu.raw(),
      1,
      offsets.as_ptr(),
      term_ids.as_ptr(),
      weights.as_ptr(),
      &plans,
      &filter_id,
      k,
      strategy,
    )
  };
  if batch.is_null() {
    return Err(last_error());
  }
  let (mut doc, mut seg, mut score) = (vec![0u32; top_k], vec![0u32; top_k], vec![0f32; top_k]);
  let mut count = 0u32;
  let mut stats = ffi::slg_stats { scored_docs: 0, candidates_examined: 0, postings_advanced: 0 };
  let rc = unsafe {
    let mut rc = ffi::slg_batch_run(batch);
    if rc == 0 {
      rc = ffi::slg_batch_fetch(
        batch,
        doc.as_mut_ptr(),
        seg.as_mut_ptr(),
        score.as_mut_ptr(),
        &mut count,
        &mut stats,
      );
    }
    ffi::slg_batch_destroy(batch);
    rc
  };
  if rc != 0 {
    bail!("searchlite_gpu returned {rc}");
  }
  let hits = (0..count as usize).map(|i| (seg[i], doc[i] as DocId, score[i])).collect();
  Ok((hits, stats.scored_docs))
}
