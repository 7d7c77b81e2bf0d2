// gp_bc.h -- device-resident state shared by stage B (stage_b.hip) and stage C
// (stage_c.hip): the per-contig sequence slots and the ntEdit node lists.
#pragma once

#include "gp_internal.h"

namespace gp {

struct Node {
    int32_t type; // -1 unset, 0 = contig[s..e], 1 = inserted char c
    uint32_t s, e;
    uint32_t c;
};

struct StageBC {
    // two sequence buffers with identical per-contig slots; cur flips per pass
    DevBuf<uint8_t> seq[2];
    DevBuf<uint32_t> len[2];
    DevBuf<uint64_t> off;     // [n_contigs+1] slot offsets (bytes)
    DevBuf<uint8_t> dropped;  // [n_contigs]
    DevBuf<uint64_t> pack_off; // [n_contigs + 1] output offsets of the export
    DevBuf<uint8_t> pack_buf;  // staging for the host export
    DevBuf<Node> nodes;
    DevBuf<uint64_t> node_off; // [n_contigs+1]
    DevBuf<uint32_t> contig_batch, name_len, in_len;
    DevBuf<uint64_t> in_off;
    DevBuf<uint32_t> batch_begin; // [n_batches+1]
    DevBuf<uint32_t> err;     // device-side error word
    DevBuf<unsigned long long> stats; // b_scan_kmers, b_sites, b_probes, b_edits, c_*
    DevBuf<uint8_t> batch_skip;
    DevBuf<unsigned long long> bm_site, bm_ok; // stage B bitmaps, one bit per slot position
    DevBuf<uint4> site_rec;    // one SiteRec (16 B) per slot position, see ntedit_site_eval; empty = disabled
    DevBuf<uint32_t> n_nodes;           // [n_contigs]
    DevBuf<uint32_t> flags;             // [n_contigs] classify_kernel: 1 = not plain ACGT, 2 = has a gap char
    DevBuf<uint32_t> tile_contig, tile_first; // output tiles of materialize_kernel
    uint32_t n_tiles = 0;
    // long contigs walked as segments (stage_b.hip, SegDesc): per-pass descriptors and results
    DevBuf<uint8_t> seg_desc, seg_out, seg_contigs, seg_req, seg_req2, seg_state; // SegDesc[], SegOut[], SegContig[], BoundaryReq[]
    DevBuf<uint32_t> seg_cut, contig_mode, seg_order;
    std::vector<uint64_t> h_off, h_node_off;
    int cur = 0;
    uint64_t total_slots = 0;
    // stage C buffers live in stage_c.hip
    void* c_state = nullptr;
};


// tile-parallel helpers of stage_b.hip that stage C reuses
int launch_classify(gp_job* job);
int launch_copy_flagged(gp_job* job, uint32_t reject, uint32_t min_len = 0);
int stage_c_prepare(gp_job* job);
void stage_c_release(gp_job* job);

} // namespace gp
