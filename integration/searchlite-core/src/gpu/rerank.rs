//! searchlite-core/src/gpu/rerank.rs — replaces the identity stub `rerank(entries) = entries.to_vec()`
//! (gpu/rerank.rs:1-5 of the surveyed snapshot) by the device rerank: vector similarity of the
//! candidates against the query vector(s), alpha-blended with their BM25 scores exactly as
//! compute_hybrid_score does (api/reader.rs:225-254), top-`k_out` returned.
//!
//! UNVERIFIED SOURCE (never compiled: no Rust toolchain in the GPU library's build image).

use anyhow::{bail, Result};

use super::{ffi, GpuSegments};
use crate::DocId;

/// One vector clause of a request (VectorClausePlan, api/reader.rs:146-155): the query vector is
/// already normalized for cosine (api/reader.rs:2114-2120).
pub struct RerankClause<'a> {
  pub vector: &'a [f32],
  pub alpha: f32,
  pub boost: f32,
}

/// entries: (segment_ord, doc_id, bm25) of the BM25 pass in any order; returns
/// (segment_ord, doc_id, blended, vector_sum) sorted by (blended desc, segment asc, doc asc).
/// Natural call site: right after the cross-segment sort where rescore_hits runs today
/// (api/reader.rs:2776-2795), with the clauses of build_vector_plan (api/reader.rs:2001-2183).
pub fn rerank(
  gpu: &GpuSegments,
  entries: &[(u32, DocId, f32)],
  clauses: &[RerankClause<'_>],
  k_out: usize,
) -> Result<Vec<(u32, DocId, f32, f32)>> {
  if clauses.is_empty() || clauses.len() > 8 {
    bail!("1..=8 vector clauses (MAX_VECTOR_CLAUSES, api/reader.rs:134)");
  }
  let dim = clauses[0].vector.len();
  let n = entries.len();
  let mut qvecs = Vec::with_capacity(dim * clauses.len());
  for c in clauses {
    if c.vector.len() != dim {
      bail!("vector clauses of different dimensions");
    }
    qvecs.extend_from_slice(c.vector);
  }
  let alpha: Vec<f32> = clauses.iter().map(|c| c.alpha).collect();
  let boost: Vec<f32> = clauses.iter().map(|c| c.boost).collect();
  let seg: Vec<u32> = entries.iter().map(|e| e.0).collect();
  let doc: Vec<u32> = entries.iter().map(|e| e.1 as u32).collect();
  let bm25: Vec<f32> = entries.iter().map(|e| e.2).collect();
  let count = n as u32;
  let (mut od, mut os) = (vec![0u32; k_out], vec![0u32; k_out]);
  let (mut osc, mut ov) = (vec![0f32; k_out], vec![0f32; k_out]);
  let mut oc = 0u32;
  let rc = unsafe {
    ffi::slg_rerank_multi_batch(
      gpu.raw(),
      1,
      clauses.len() as u32,
      qvecs.as_ptr(),
      alpha.as_ptr(),
      boost.as_ptr(),
      doc.as_ptr(),
      seg.as_ptr(),
      bm25.as_ptr(),
      &count,
      n as u32,
      k_out as u32,
      od.as_mut_ptr(),
      os.as_mut_ptr(),
      osc.as_mut_ptr(),
      ov.as_mut_ptr(),
      &mut oc,
    )
  };
  if rc != 0 {
    bail!("searchlite_gpu rerank returned {rc}");
  }
  Ok((0..oc as usize).map(|i| (os[i], od[i] as DocId, osc[i], ov[i])).collect())
}
