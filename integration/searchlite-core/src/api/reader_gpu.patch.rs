// integration/searchlite-core/src/api/reader_gpu.patch.rs — the lines a maintainer adds to
// IndexReader (searchlite-core/src/api/reader.rs of the surveyed snapshot).  UNVERIFIED SOURCE.
//
// (1) The staged index lives on the INDEX, not on the reader: searchlite opens a fresh IndexReader
//     per request (searchlite-http/src/lib.rs:640-643 -> index/mod.rs:98-100 ->
//     api/reader.rs:1887-1913), and staging per reader would upload the whole index per request.
//
//     InnerIndex (index/mod.rs) gains the cache slot:
//
//       #[cfg(feature = "gpu")]
//       pub(crate) gpu: std::sync::Mutex<Option<std::sync::Arc<crate::gpu::GpuSegments>>>,
//
//     IndexReader gains a field and fills it in IndexReader::open, after `segments`:
//
//       #[cfg(feature = "gpu")]
//       gpu: Option<std::sync::Arc<crate::gpu::GpuSegments>>,
//
//       #[cfg(feature = "gpu")]
//       let gpu = {
//         let schema = &manifest.schema;
//         let fields: Vec<String> = schema.text_fields.iter().map(|f| f.name.clone())
//           .chain(schema.keyword_fields.iter().map(|f| f.name.clone())).collect();
//         let vfield = schema.vector_fields.first().map(|f| f.name.as_str());
//         // unchanged manifest: an Arc clone.  After a commit: the device index follows it in place
//         // (slg_index_update_deleted / _add_segment / _remove_segment); a missing GPU / library is
//         // not an error: the CPU scorer serves everything
//         crate::gpu::GpuSegments::for_reader(&inner.gpu, &segments, &fields, vfield,
//                                            options.bm25_k1, options.bm25_b, 0)
//       };
//
// (2) inside IndexReader::search, right after `expand_term_groups` (api/reader.rs:2629-2635) and the
//     `root_filter` binding (:2654-2658), BEFORE the `for (segment_ord, seg) in self.segments` loop
//     (:2670).  When the block below produces `hits`, the loop is skipped; everything after it
//     (vector merge :2754-2775, sort :2776-2778 — a no-op on already ordered hits —, rescore,
//     truncation to `limit`, cursor encoding, materialize_hit) runs unchanged.

#[cfg(feature = "gpu")]
let gpu_hits: Option<Vec<RankedHit>> = (|| {
  let gpu = self.gpu.as_ref()?;
  let folded = crate::gpu::fold_terms(
    qualified_terms.iter().map(|t| (t.key.as_str(), t.weight, t.leaf)),
  );
  let (score_plan, n_leaves, min_match) = crate::gpu::gpu_eligible(
    req, &sort_plan, &query_plan, needs_score_hook, top_k, folded.len(),
  )?;
  // term keys of the matcher's not-term groups (api/reader.rs:1499-1503): rejected on the device by a filter
  // built from their posting lists
  let not_keys: Vec<String> = match &query_plan.matcher {
    QueryMatcher::QueryString(qs) => qs
      .not_term_groups
      .iter()
      .filter_map(|i| term_groups.get(*i))
      .flat_map(|g| g.keys.iter().cloned())
      .collect(),
    _ => Vec::new(),
  };
  match crate::gpu::gpu_top_k(
    gpu, &self.segments, &folded, &score_plan, n_leaves, min_match, req.filter.as_ref(), &not_keys,
    &req.execution, top_k,
  ) {
    Ok((rows, scored)) => {
      // total_hits_estimate: the CPU path counts the docs `accept` saw (pruning-dependent under
      // Wand/Bmw, api/reader.rs:3029-3031); the device reports every distinct doc it scored
      total_matches = scored;
      Some(
        rows
          .into_iter()
          .map(|(segment_ord, doc_id, score)| RankedHit {
            key: sort_plan.build_key(&self.segments[segment_ord as usize], doc_id, score, segment_ord),
            score,
            vector_score: None,
            explanation: None,
          })
          .collect(),
      )
    }
    Err(_) => None, // any library error: fall back to the CPU scorer below
  }
})();
#[cfg(feature = "gpu")]
let skip_cpu_segments = gpu_hits.is_some();
#[cfg(not(feature = "gpu"))]
let skip_cpu_segments = false;
#[cfg(feature = "gpu")]
if let Some(h) = gpu_hits {
  hits = h;
}
// for (segment_ord, seg) in self.segments.iter().enumerate() {   // :2670
//   if skip_cpu_segments { break; }
//   ...
