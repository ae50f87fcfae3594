//! GPU-backed drop-in for trueno_rag::index::VectorStore over libtrag.so (include/trag.h).
//! UNCOMPILED: the image that produced this repository has no Rust toolchain.
// trueno-rag-hip/src/lib.rs
#![allow(unsafe_code)]
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_void};
use trueno_rag::{Chunk, ChunkId, Error, Result};
use trueno_rag::index::{DistanceMetric, VectorStoreConfig};

#[repr(C)]
struct TragConfig { dim: u32, metric: u32, device: i32, scan_mode: u32, reserve_rows: u64 }
#[repr(C)] pub struct TragStore { _p: [u8; 0] }

extern "C" {
    fn trag_store_create(cfg: *const TragConfig, out: *mut *mut TragStore) -> i32;
    fn trag_store_destroy(s: *mut TragStore);
    fn trag_store_insert(s: *mut TragStore, rows: *const f32, n: u64, dim: u32, first: *mut u32) -> i32;
    fn trag_store_overwrite(s: *mut TragStore, row: u32, v: *const f32, dim: u32) -> i32;
    fn trag_store_remove(s: *mut TragStore, row: u32) -> i32;
    fn trag_store_len(s: *const TragStore) -> u64;
    fn trag_store_set_bf16_mirror(s: *mut TragStore, enabled: i32) -> i32;
    fn trag_store_search(s: *mut TragStore, q: *const f32, dim: u32, k: u32,
                         rows: *mut u32, scores: *mut f32, n: *mut u32) -> i32;
    fn trag_store_search_batch(s: *mut TragStore, q: *const f32, nq: u32, dim: u32, k: u32,
                               rows: *mut u32, scores: *mut f32, n: *mut u32) -> i32;
    fn trag_fuse(strategy: u32, param: f32, d_ids: *const u64, d_sc: *const f32, nd: u32,
                 s_ids: *const u64, s_sc: *const f32, ns: u32,
                 out_ids: *mut u64, out_sc: *mut f32, out_n: *mut u32) -> i32;
    fn trag_last_error() -> *const c_char;
}

fn err(code: i32, expected: usize, actual: usize) -> Error {
    let msg = unsafe { CStr::from_ptr(trag_last_error()) }.to_string_lossy().into_owned();
    match code {
        1 => Error::DimensionMismatch { expected, actual },      // src/index.rs:365-370, 387-392
        2 => Error::InvalidConfig(msg),                          // src/index.rs:360-363
        _ => Error::VectorStore(msg),                            // src/error.rs:38-39
    }
}

/// Same method surface as trueno_rag::index::VectorStore (src/index.rs:332-437).
pub struct GpuVectorStore {
    config: VectorStoreConfig,
    h: *mut TragStore,
    row_of: HashMap<ChunkId, u32>,       // id -> row
    id_of: Vec<Option<ChunkId>>,         // row -> id (None after remove)
    chunks: HashMap<ChunkId, Chunk>,     // `chunks` of the reference, src/index.rs:329
}
unsafe impl Send for GpuVectorStore {}   // the handle serialises searches internally
unsafe impl Sync for GpuVectorStore {}

impl GpuVectorStore {
    pub fn new(config: VectorStoreConfig) -> Result<Self> {
        let metric = match config.metric { DistanceMetric::Cosine => 0, DistanceMetric::Euclidean => 1,
                                           DistanceMetric::DotProduct => 2 };
        let c = TragConfig { dim: config.dimension as u32, metric, device: -1, scan_mode: 0, reserve_rows: 0 };
        let mut h = std::ptr::null_mut();
        let rc = unsafe { trag_store_create(&c, &mut h) };
        if rc != 0 { return Err(err(rc, 0, 0)); }
        Ok(Self { config, h, row_of: HashMap::new(), id_of: Vec::new(), chunks: HashMap::new() })
    }
    pub fn with_dimension(dimension: usize) -> Result<Self> {
        Self::new(VectorStoreConfig { dimension, ..Default::default() })
    }
    pub fn config(&self) -> &VectorStoreConfig { &self.config }

    pub fn insert(&mut self, chunk: Chunk) -> Result<()> {
        let e = chunk.embedding.as_ref()
            .ok_or_else(|| Error::InvalidConfig("chunk must have embedding".to_string()))?;
        let dim = self.config.dimension;
        if e.len() != dim { return Err(Error::DimensionMismatch { expected: dim, actual: e.len() }); }
        if let Some(&row) = self.row_of.get(&chunk.id) {          // duplicate id overwrites, :372
            let rc = unsafe { trag_store_overwrite(self.h, row, e.as_ptr(), dim as u32) };
            if rc != 0 { return Err(err(rc, dim, e.len())); }
        } else {
            let mut first = 0u32;
            let rc = unsafe { trag_store_insert(self.h, e.as_ptr(), 1, dim as u32, &mut first) };
            if rc != 0 { return Err(err(rc, dim, e.len())); }
            self.row_of.insert(chunk.id, first);
            if self.id_of.len() <= first as usize { self.id_of.resize(first as usize + 1, None); }
            self.id_of[first as usize] = Some(chunk.id);
        }
        self.chunks.insert(chunk.id, chunk);
        Ok(())
    }
    pub fn insert_batch(&mut self, chunks: Vec<Chunk>) -> Result<()> {
        for c in chunks { self.insert(c)?; }                       // a bulk path can pass n rows at once
        Ok(())
    }
    pub fn search(&self, query_vector: &[f32], k: usize) -> Result<Vec<(ChunkId, f32)>> {
        let dim = self.config.dimension;
        if query_vector.len() != dim {
            return Err(Error::DimensionMismatch { expected: dim, actual: query_vector.len() });
        }
        let (mut rows, mut scores, mut n) = (vec![0u32; k.max(1)], vec![0f32; k.max(1)], 0u32);
        let rc = unsafe { trag_store_search(self.h, query_vector.as_ptr(), dim as u32, k as u32,
                                            rows.as_mut_ptr(), scores.as_mut_ptr(), &mut n) };
        if rc != 0 { return Err(err(rc, dim, query_vector.len())); }
        Ok((0..n as usize).map(|i| (self.id_of[rows[i] as usize].unwrap(), scores[i])).collect())
    }
    pub fn get(&self, id: ChunkId) -> Option<&Chunk> { self.chunks.get(&id) }
    pub fn remove(&mut self, id: ChunkId) -> Option<Chunk> {
        let row = self.row_of.remove(&id)?;
        unsafe { trag_store_remove(self.h, row) };
        self.id_of[row as usize] = None;
        self.chunks.remove(&id)
    }
    pub fn len(&self) -> usize { unsafe { trag_store_len(self.h) as usize } }
    pub fn is_empty(&self) -> bool { self.len() == 0 }
    /// Not in the reference: a bf16 copy of the corpus in HBM (1.5 x the memory) for the
    /// prefiltering passes; search results stay the exact f32 results.
    pub fn set_bf16_mirror(&mut self, enabled: bool) -> Result<()> {
        let rc = unsafe { trag_store_set_bf16_mirror(self.h, enabled as i32) };
        if rc != 0 { Err(err(rc, 0, 0)) } else { Ok(()) }
    }
}
impl Drop for GpuVectorStore { fn drop(&mut self) { unsafe { trag_store_destroy(self.h) } } }

// ---------------------------------------------------------------------------
// BM25: `impl SparseIndex for GpuBm25Index` (src/index.rs:8-28, 175-280).
// The tokenizer is the reference's own (`BM25Index::tokenize`, a pub fn,
// src/index.rs:111-124): a default `BM25Index` is kept only to call it.
// ---------------------------------------------------------------------------
#[repr(C)]
struct TragBm25Config { k1: f32, b: f32, device: i32 }
#[repr(C)] pub struct TragBm25 { _p: [u8; 0] }

extern "C" {
    fn trag_bm25_create(cfg: *const TragBm25Config, out: *mut *mut TragBm25) -> i32;
    fn trag_bm25_destroy(ix: *mut TragBm25);
    fn trag_bm25_add(ix: *mut TragBm25, terms: *const u32, doc_offsets: *const u64, n_docs: u64,
                     first: *mut u32) -> i32;
    fn trag_bm25_remove(ix: *mut TragBm25, doc: u32) -> i32;
    fn trag_bm25_len(ix: *const TragBm25) -> u64;
    fn trag_bm25_search(ix: *mut TragBm25, q: *const u32, n_terms: u32, k: u32,
                        docs: *mut u32, scores: *mut f32, n: *mut u32) -> i32;
}

pub struct GpuBm25Index {
    h: *mut TragBm25,
    tokenizer: trueno_rag::index::BM25Index,   // never indexed into; tokenize() only
    terms: HashMap<String, u32>,               // term -> id
    docs_of: HashMap<ChunkId, Vec<u32>>,       // a re-added id owns several documents
    id_of: Vec<ChunkId>,
}
unsafe impl Send for GpuBm25Index {}
unsafe impl Sync for GpuBm25Index {}

impl GpuBm25Index {
    pub fn new() -> Self { Self::with_params(1.2, 0.75) }
    pub fn with_params(k1: f32, b: f32) -> Self {
        let c = TragBm25Config { k1, b, device: -1 };
        let mut h = std::ptr::null_mut();
        let rc = unsafe { trag_bm25_create(&c, &mut h) };
        assert_eq!(rc, 0, "trag_bm25_create: {:?}", err(rc, 0, 0));   // `new` is infallible in the reference
        Self { h, tokenizer: trueno_rag::index::BM25Index::with_params(k1, b), terms: HashMap::new(),
               docs_of: HashMap::new(), id_of: Vec::new() }
    }
    fn ids(&mut self, tokens: Vec<String>, create: bool) -> Vec<u32> {
        tokens.into_iter().map(|t| {
            if let Some(&i) = self.terms.get(&t) { i }
            else if create { let i = self.terms.len() as u32; self.terms.insert(t, i); i }
            else { 0x0FFF_FFFF }                                      // never assigned: matches nothing
        }).collect()
    }
}

impl trueno_rag::index::SparseIndex for GpuBm25Index {
    fn add(&mut self, chunk: &Chunk) {
        let toks = self.tokenizer.tokenize(&chunk.content);
        let ids = self.ids(toks, true);
        let off = [0u64, ids.len() as u64];
        let mut first = 0u32;
        let p = if ids.is_empty() { [0u32].as_ptr() } else { ids.as_ptr() };
        let rc = unsafe { trag_bm25_add(self.h, p, off.as_ptr(), 1, &mut first) };
        assert_eq!(rc, 0, "trag_bm25_add: {:?}", err(rc, 0, 0));
        self.docs_of.entry(chunk.id).or_default().push(first);
        self.id_of.push(chunk.id);
    }
    fn add_batch(&mut self, chunks: &[Chunk]) { for c in chunks { self.add(c); } }   // or one call, n_docs = len
    fn search(&self, query: &str, k: usize) -> Vec<(ChunkId, f32)> {
        let toks = self.tokenizer.tokenize(query);
        if toks.is_empty() || k == 0 { return Vec::new(); }
        let ids: Vec<u32> = toks.iter().map(|t| *self.terms.get(t).unwrap_or(&0x0FFF_FFFF)).collect();
        let (mut docs, mut scores, mut n) = (vec![0u32; k], vec![0f32; k], 0u32);
        let rc = unsafe { trag_bm25_search(self.h, ids.as_ptr(), ids.len() as u32, k as u32,
                                           docs.as_mut_ptr(), scores.as_mut_ptr(), &mut n) };
        if rc != 0 { return Vec::new(); }                             // the trait has no error channel
        (0..n as usize).map(|i| (self.id_of[docs[i] as usize], scores[i])).collect()
    }
    fn remove(&mut self, chunk_id: ChunkId) {
        for d in self.docs_of.remove(&chunk_id).unwrap_or_default() { unsafe { trag_bm25_remove(self.h, d) }; }
    }
    fn len(&self) -> usize { unsafe { trag_bm25_len(self.h) as usize } }
}
impl Drop for GpuBm25Index { fn drop(&mut self) { unsafe { trag_bm25_destroy(self.h) } } }
