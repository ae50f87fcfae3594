//! GPU-backed drop-in for trueno_rag::index::VectorStore over libtrag.so (include/trag.h).
//! UNCOMPILED: the image that produced this repository has no Rust toolchain.
//!
//! With integration/denseindex.patch applied to trueno-rag, `GpuVectorStore` and
//! `GpuShardedVectorStore` implement `trueno_rag::DenseIndex` and `GpuBm25Index` the crate's own
//! `SparseIndex`, so `HybridRetriever::new(gpu_store, gpu_bm25, embedder)` and
//! `RagPipelineBuilder::new().dense_index(gpu_store)` take them as they are.
// trueno-rag-hip/src/lib.rs
#![allow(unsafe_code)]
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_void};
use trueno_rag::{Chunk, ChunkId, DenseIndex, Error, Result};
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
    fn trag_store_set_pipelined(s: *mut TragStore, enabled: i32) -> i32;
    fn trag_store_fence(s: *mut TragStore, hip_stream: *mut c_void) -> i32;
    fn trag_store_len(s: *const TragStore) -> u64;
    fn trag_store_rows(s: *const TragStore) -> u64;
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

// One process, several GPUs: a store striped over the devices, searched with one call
// (include/trag.h, trag_sharded_*). `exchange`: 0 = RCCL broadcast / all-gather, 1 = copies.
#[repr(C)] pub struct TragShardedStore { _p: [u8; 0] }
extern "C" {
    fn trag_sharded_create(cfg: *const TragConfig, n_devices: u32, device_ids: *const i32, exchange: u32,
                           out: *mut *mut TragShardedStore) -> i32;
    fn trag_sharded_destroy(s: *mut TragShardedStore);
    fn trag_sharded_insert(s: *mut TragShardedStore, rows: *const f32, n: u64, dim: u32, first: *mut u32) -> i32;
    fn trag_sharded_overwrite(s: *mut TragShardedStore, row: u32, v: *const f32, dim: u32) -> i32;
    fn trag_sharded_remove(s: *mut TragShardedStore, row: u32) -> i32;
    fn trag_sharded_len(s: *const TragShardedStore) -> u64;
    fn trag_sharded_search(s: *mut TragShardedStore, q: *const f32, dim: u32, k: u32,
                           rows: *mut u32, scores: *mut f32, n: *mut u32) -> i32;
    fn trag_sharded_search_batch(s: *mut TragShardedStore, q: *const f32, nq: u32, dim: u32, k: u32,
                                 rows: *mut u32, scores: *mut f32, n: *mut u32) -> i32;
}

// One process per GPU (torchrun / MPI style): every rank holds its shard in a `trag_store` and
// joins a group; `trag_shard_search_batch` is collective (same queries on every rank) and returns
// the global top k on every rank. The 128-byte id comes from rank 0 (`trag_shard_unique_id`) and
// reaches the other ranks by whatever channel the launcher has.
#[repr(C)] pub struct TragShardGroup { _p: [u8; 0] }
#[repr(C)]
pub struct TragShardConfig { pub rank: u32, pub world: u32, pub device: i32, pub k_max: u32,
                             pub max_queries: u32, pub overlap: u32 }
extern "C" {
    pub fn trag_shard_unique_id(id: *mut u8) -> i32;
    pub fn trag_shard_group_create(cfg: *const TragShardConfig, id: *const u8, out: *mut *mut TragShardGroup) -> i32;
    pub fn trag_shard_group_destroy(g: *mut TragShardGroup);
    pub fn trag_shard_search_batch(g: *mut TragShardGroup, local: *mut TragStore, row0: u32, row_mul: u32,
                                   q: *const f32, nq: u32, dim: u32, k: u32,
                                   rows: *mut u32, scores: *mut f32, n: *mut u32) -> i32;
    pub fn trag_shard_search_device(g: *mut TragShardGroup, local: *mut TragStore, row0: u32, row_mul: u32,
                                    d_q: *const f32, nq: u32, dim: u32, k: u32, d_rows: *mut u32,
                                    d_scores: *mut f32, d_n: *mut u32, d_status: *mut u32,
                                    hip_stream: *mut c_void) -> i32;
    pub fn trag_shard_fence(g: *mut TragShardGroup, hip_stream: *mut c_void) -> i32;
    pub fn trag_shard_rccl_version(version: *mut i32) -> i32;
    pub fn trag_shard_group_info(g: *mut TragShardGroup, comm_world: *mut u32, comm_rank: *mut u32,
                                 comm_device: *mut i32) -> i32;
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

fn metric_code(m: DistanceMetric) -> u32 {
    match m { DistanceMetric::Cosine => 0, DistanceMetric::Euclidean => 1, DistanceMetric::DotProduct => 2 }
}

impl GpuVectorStore {
    /// Fallible constructor: no HIP device, device out of memory, dimension 0.
    pub fn try_new(config: VectorStoreConfig) -> Result<Self> {
        let c = TragConfig { dim: config.dimension as u32, metric: metric_code(config.metric), device: -1,
                             scan_mode: 0, reserve_rows: 0 };
        let mut h = std::ptr::null_mut();
        let rc = unsafe { trag_store_create(&c, &mut h) };
        if rc != 0 { return Err(err(rc, 0, 0)); }
        Ok(Self { config, h, row_of: HashMap::new(), id_of: Vec::new(), chunks: HashMap::new() })
    }
    /// Infallible like `VectorStore::new` (src/index.rs:335-343): a machine without a usable GPU is
    /// a deployment error, reported by a panic that carries the library's message.
    pub fn new(config: VectorStoreConfig) -> Self {
        Self::try_new(config).unwrap_or_else(|e| panic!("GpuVectorStore::new: {e}"))
    }
    pub fn with_dimension(dimension: usize) -> Self {
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
    /// One FFI call for the new rows of the batch (ids already present are overwritten in place,
    /// one call each, as `HashMap::insert` would; an id that occurs twice in the batch keeps its
    /// last embedding). As in the reference, a chunk without an embedding or of the wrong dimension
    /// fails the call; unlike it, nothing of the batch has been inserted then.
    pub fn insert_batch(&mut self, chunks: Vec<Chunk>) -> Result<()> {
        let dim = self.config.dimension;
        for c in &chunks {
            let e = c.embedding.as_ref()
                .ok_or_else(|| Error::InvalidConfig("chunk must have embedding".to_string()))?;
            if e.len() != dim { return Err(Error::DimensionMismatch { expected: dim, actual: e.len() }); }
        }
        let mut fresh: Vec<Chunk> = Vec::new();
        let mut slot: HashMap<ChunkId, usize> = HashMap::new();      // id -> position in `fresh`
        for c in chunks {
            if self.row_of.contains_key(&c.id) { self.insert(c)?; }
            else if let Some(&i) = slot.get(&c.id) { fresh[i] = c; }
            else { slot.insert(c.id, fresh.len()); fresh.push(c); }
        }
        if fresh.is_empty() { return Ok(()); }
        let mut flat: Vec<f32> = Vec::with_capacity(fresh.len() * dim);
        for c in &fresh { flat.extend_from_slice(c.embedding.as_ref().unwrap()); }
        let mut first = 0u32;
        let rc = unsafe { trag_store_insert(self.h, flat.as_ptr(), fresh.len() as u64, dim as u32, &mut first) };
        if rc != 0 { return Err(err(rc, dim, dim)); }
        let end = first as usize + fresh.len();
        if self.id_of.len() < end { self.id_of.resize(end, None); }
        for (i, c) in fresh.into_iter().enumerate() {
            self.row_of.insert(c.id, first + i as u32);
            self.id_of[first as usize + i] = Some(c.id);
            self.chunks.insert(c.id, c);
        }
        Ok(())
    }
    pub fn search(&self, query_vector: &[f32], k: usize) -> Result<Vec<(ChunkId, f32)>> {
        let dim = self.config.dimension;
        if query_vector.len() != dim {
            return Err(Error::DimensionMismatch { expected: dim, actual: query_vector.len() });
        }
        let k = k.min(self.len());                                 // the reference truncates: any k is legal
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
    /// Not in the reference: for callers of the device entry (queries and results in device memory, the
    /// raw handle from `handle()`). Mode 2 defers the select of a single query into the launch of the
    /// next query's scan, mode 1 runs it on a side stream of the handle, 0 switches both off.
    pub fn set_pipelined(&mut self, mode: i32) -> Result<()> {
        let rc = unsafe { trag_store_set_pipelined(self.h, mode) };
        if rc != 0 { Err(err(rc, 0, 0)) } else { Ok(()) }
    }
    /// Results of pipelined device-entry searches may be read on `hip_stream` only behind this (in mode 2
    /// the last query's select is not even launched before it).
    pub fn fence(&self, hip_stream: *mut c_void) -> Result<()> {
        let rc = unsafe { trag_store_fence(self.h, hip_stream) };
        if rc != 0 { Err(err(rc, 0, 0)) } else { Ok(()) }
    }
    pub fn handle(&self) -> *mut TragStore { self.h }
}
impl Drop for GpuVectorStore { fn drop(&mut self) { unsafe { trag_store_destroy(self.h) } } }

// `Clone` / `Debug` of the reference's store (src/index.rs:322). A clone is a second store on the
// same device with the same rows (read back from HBM and inserted again): the two are independent
// afterwards, like two copies of the reference's HashMaps.
impl Clone for GpuVectorStore {
    fn clone(&self) -> Self {
        let mut other = Self::new(self.config.clone());
        let mut live: Vec<Chunk> = Vec::with_capacity(self.chunks.len());
        for id in self.id_of.iter().flatten() {                     // row order: the clone's rows stay compact
            if let Some(c) = self.chunks.get(id) { live.push(c.clone()); }
        }
        other.insert_batch(live).unwrap_or_else(|e| panic!("GpuVectorStore::clone: {e}"));
        other
    }
}
impl std::fmt::Debug for GpuVectorStore {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("GpuVectorStore").field("config", &self.config).field("len", &self.len())
            .field("rows", &unsafe { trag_store_rows(self.h) }).finish()
    }
}

impl DenseIndex for GpuVectorStore {
    fn insert(&mut self, chunk: Chunk) -> Result<()> { GpuVectorStore::insert(self, chunk) }
    fn insert_batch(&mut self, chunks: Vec<Chunk>) -> Result<()> { GpuVectorStore::insert_batch(self, chunks) }
    fn search(&self, q: &[f32], k: usize) -> Result<Vec<(ChunkId, f32)>> { GpuVectorStore::search(self, q, k) }
    fn get(&self, id: ChunkId) -> Option<&Chunk> { GpuVectorStore::get(self, id) }
    fn remove(&mut self, id: ChunkId) -> Option<Chunk> { GpuVectorStore::remove(self, id) }
    fn len(&self) -> usize { GpuVectorStore::len(self) }
}

/// The same store striped over several GPUs of one process (global row g on device g % n): one
/// search call scans every shard and merges on the library's streams (src/retrieve.rs:181 is
/// the call it serves). Same surface and id bookkeeping as `GpuVectorStore`.
pub struct GpuShardedVectorStore {
    config: VectorStoreConfig,
    h: *mut TragShardedStore,
    row_of: HashMap<ChunkId, u32>,
    id_of: Vec<Option<ChunkId>>,
    chunks: HashMap<ChunkId, Chunk>,
}
unsafe impl Send for GpuShardedVectorStore {}
unsafe impl Sync for GpuShardedVectorStore {}

impl GpuShardedVectorStore {
    /// `devices`: HIP ordinals, one shard each. `use_rccl`: exchange by RCCL collectives (false:
    /// device-to-device copies ordered by events).
    pub fn try_new(config: VectorStoreConfig, devices: &[i32], use_rccl: bool) -> Result<Self> {
        let c = TragConfig { dim: config.dimension as u32, metric: metric_code(config.metric), device: -1,
                             scan_mode: 0, reserve_rows: 0 };
        let mut h = std::ptr::null_mut();
        let rc = unsafe { trag_sharded_create(&c, devices.len() as u32, devices.as_ptr(),
                                              if use_rccl { 0 } else { 1 }, &mut h) };
        if rc != 0 { return Err(err(rc, 0, 0)); }
        Ok(Self { config, h, row_of: HashMap::new(), id_of: Vec::new(), chunks: HashMap::new() })
    }
    /// Up to 4096 / n_devices results per query (the merge's record limit).
    pub fn search_batch(&self, queries: &[f32], k: usize) -> Result<Vec<Vec<(ChunkId, f32)>>> {
        let dim = self.config.dimension;
        if dim == 0 || queries.len() % dim != 0 {
            return Err(Error::DimensionMismatch { expected: dim, actual: queries.len() });
        }
        let nq = queries.len() / dim;
        let k = k.min(DenseIndex::len(self));
        let (mut rows, mut scores, mut n) = (vec![0u32; (nq * k).max(1)], vec![0f32; (nq * k).max(1)], vec![0u32; nq.max(1)]);
        let rc = unsafe { trag_sharded_search_batch(self.h, queries.as_ptr(), nq as u32, dim as u32, k as u32,
                                                    rows.as_mut_ptr(), scores.as_mut_ptr(), n.as_mut_ptr()) };
        if rc != 0 { return Err(err(rc, dim, dim)); }
        Ok((0..nq).map(|b| (0..n[b] as usize)
            .map(|i| (self.id_of[rows[b * k + i] as usize].unwrap(), scores[b * k + i])).collect()).collect())
    }
}
impl Drop for GpuShardedVectorStore { fn drop(&mut self) { unsafe { trag_sharded_destroy(self.h) } } }

impl DenseIndex for GpuShardedVectorStore {
    fn insert(&mut self, chunk: Chunk) -> Result<()> {
        let e = chunk.embedding.as_ref()
            .ok_or_else(|| Error::InvalidConfig("chunk must have embedding".to_string()))?;
        let dim = self.config.dimension;
        if e.len() != dim { return Err(Error::DimensionMismatch { expected: dim, actual: e.len() }); }
        if let Some(&row) = self.row_of.get(&chunk.id) {
            let rc = unsafe { trag_sharded_overwrite(self.h, row, e.as_ptr(), dim as u32) };
            if rc != 0 { return Err(err(rc, dim, e.len())); }
        } else {
            let mut first = 0u32;
            let rc = unsafe { trag_sharded_insert(self.h, e.as_ptr(), 1, dim as u32, &mut first) };
            if rc != 0 { return Err(err(rc, dim, e.len())); }
            self.row_of.insert(chunk.id, first);
            if self.id_of.len() <= first as usize { self.id_of.resize(first as usize + 1, None); }
            self.id_of[first as usize] = Some(chunk.id);
        }
        self.chunks.insert(chunk.id, chunk);
        Ok(())
    }
    fn search(&self, query_vector: &[f32], k: usize) -> Result<Vec<(ChunkId, f32)>> {
        let dim = self.config.dimension;
        if query_vector.len() != dim {
            return Err(Error::DimensionMismatch { expected: dim, actual: query_vector.len() });
        }
        let k = k.min(DenseIndex::len(self));
        let (mut rows, mut scores, mut n) = (vec![0u32; k.max(1)], vec![0f32; k.max(1)], 0u32);
        let rc = unsafe { trag_sharded_search(self.h, query_vector.as_ptr(), dim as u32, k as u32,
                                              rows.as_mut_ptr(), scores.as_mut_ptr(), &mut n) };
        if rc != 0 { return Err(err(rc, dim, query_vector.len())); }
        Ok((0..n as usize).map(|i| (self.id_of[rows[i] as usize].unwrap(), scores[i])).collect())
    }
    fn get(&self, id: ChunkId) -> Option<&Chunk> { self.chunks.get(&id) }
    fn remove(&mut self, id: ChunkId) -> Option<Chunk> {
        let row = self.row_of.remove(&id)?;
        unsafe { trag_sharded_remove(self.h, row) };
        self.id_of[row as usize] = None;
        self.chunks.remove(&id)
    }
    fn len(&self) -> usize { unsafe { trag_sharded_len(self.h) as usize } }
}

// ---------------------------------------------------------------------------
// BM25: `impl SparseIndex for GpuBm25Index` (src/index.rs:8-28, 175-280). With the patched crate it
// serves every holder of a sparse index: `HybridRetriever::new(dense, GpuBm25Index::new(), embedder)` and
// `SparseRetriever::with_index(GpuBm25Index::new())` (src/retrieve.rs:300-335).
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
    doc_of: HashMap<ChunkId, u32>,             // id -> its document (a re-added id gets a new one)
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
               doc_of: HashMap::new(), id_of: Vec::new() }
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
    /// Re-adding a ChunkId REPLACES its document. (The reference keeps both postings, counts the
    /// chunk twice in doc_count and df while doc_lengths holds one entry, src/index.rs:176-204;
    /// that bookkeeping is not reproduced: a chunk is one document here, so a search never returns
    /// an id twice and `fuse` never counts it twice.)
    fn add(&mut self, chunk: &Chunk) {
        if let Some(old) = self.doc_of.remove(&chunk.id) { unsafe { trag_bm25_remove(self.h, old) }; }
        let toks = self.tokenizer.tokenize(&chunk.content);
        let ids = self.ids(toks, true);
        let off = [0u64, ids.len() as u64];
        let mut first = 0u32;
        let p = if ids.is_empty() { [0u32].as_ptr() } else { ids.as_ptr() };
        let rc = unsafe { trag_bm25_add(self.h, p, off.as_ptr(), 1, &mut first) };
        assert_eq!(rc, 0, "trag_bm25_add: {:?}", err(rc, 0, 0));
        self.doc_of.insert(chunk.id, first);
        if self.id_of.len() <= first as usize { self.id_of.resize(first as usize + 1, chunk.id); }
        self.id_of[first as usize] = chunk.id;
    }
    /// One FFI call for the whole batch when no id of it is indexed already or occurs twice.
    fn add_batch(&mut self, chunks: &[Chunk]) {
        let mut seen = std::collections::HashSet::new();
        if chunks.iter().any(|c| self.doc_of.contains_key(&c.id) || !seen.insert(c.id)) {
            for c in chunks { self.add(c); }
            return;
        }
        let (mut ids, mut off) = (Vec::<u32>::new(), vec![0u64]);
        for c in chunks {
            let toks = self.tokenizer.tokenize(&c.content);
            ids.extend(self.ids(toks, true));
            off.push(ids.len() as u64);
        }
        if chunks.is_empty() { return; }
        let mut first = 0u32;
        let p = if ids.is_empty() { [0u32].as_ptr() } else { ids.as_ptr() };
        let rc = unsafe { trag_bm25_add(self.h, p, off.as_ptr(), chunks.len() as u64, &mut first) };
        assert_eq!(rc, 0, "trag_bm25_add: {:?}", err(rc, 0, 0));
        let end = first as usize + chunks.len();
        if self.id_of.len() < end { self.id_of.resize(end, chunks[0].id); }
        for (i, c) in chunks.iter().enumerate() {
            self.doc_of.insert(c.id, first + i as u32);
            self.id_of[first as usize + i] = c.id;
        }
    }
    fn search(&self, query: &str, k: usize) -> Vec<(ChunkId, f32)> {
        let toks = self.tokenizer.tokenize(query);
        // any k is legal in the reference (it truncates); the GPU select serves up to 4096 results,
        // 80 x what HybridRetriever asks for (candidates_per_source = 50, src/retrieve.rs:94)
        let k = k.min(self.len()).min(4096);
        if toks.is_empty() || k == 0 { return Vec::new(); }
        let ids: Vec<u32> = toks.iter().map(|t| *self.terms.get(t).unwrap_or(&0x0FFF_FFFF)).collect();
        let (mut docs, mut scores, mut n) = (vec![0u32; k], vec![0f32; k], 0u32);
        let rc = unsafe { trag_bm25_search(self.h, ids.as_ptr(), ids.len() as u32, k as u32,
                                           docs.as_mut_ptr(), scores.as_mut_ptr(), &mut n) };
        // the trait has no error channel, and an empty result would read as "no matches": a device
        // error must not pass for that
        assert_eq!(rc, 0, "trag_bm25_search: {:?}", err(rc, 0, 0));
        (0..n as usize).map(|i| (self.id_of[docs[i] as usize], scores[i])).collect()
    }
    fn remove(&mut self, chunk_id: ChunkId) {
        if let Some(d) = self.doc_of.remove(&chunk_id) { unsafe { trag_bm25_remove(self.h, d) }; }
    }
    fn len(&self) -> usize { unsafe { trag_bm25_len(self.h) as usize } }
}
impl Drop for GpuBm25Index { fn drop(&mut self) { unsafe { trag_bm25_destroy(self.h) } } }
