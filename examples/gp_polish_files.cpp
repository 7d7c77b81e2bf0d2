// gp_polish_files.cpp -- a C++ host program above the C ABI (include/gpcore.h): FASTA in, FASTA out.
//
// What it stands in for: one `goldpolish --mappings <file> <targets> <reads> <out>` run of the
// reference (scripts/goldpolish:72-196) with everything between "index + mappings are loaded" and
// "the reaper has collected the batches" replaced by libgpcore calls:
//     goldpolish-index x2           -> gp_seqindex_build           (src/seqindex.cpp:12-125)
//     AllMappings                   -> gp_mappings_load            (src/mappings.cpp:15-320)
//     batches of -b contigs         -> record.num % bsize == bsize-1 closes a batch (scripts/goldpolish:344-355)
//     goldpolish-targeted-bfs + goldpolish-polish-batch per batch
//                                   -> gp_job_create_from_files + gp_job_polish
//     goldpolish-reaper             -> gp_job_write_fasta (records in input order)
// It is an example of the binding INTEGRATION.md describes, not a re-implementation of the
// reference's driver (no FIFOs, no workspace, no -t token file, no --target): those stay in
// GoldPolish's own scripts.  Plain C++17, no HIP in this file; link with -lgpcore.
//
// usage: gp_polish_files [-k K]... [-b bsize] [-s reads_per_10kbp] [-x mx_per_10kbp] [-d device]
//                        [-j max_Mbp_per_job] targets.fa reads.fq mappings.(paf|sam|tsv) out.fa
#include "../include/gpcore.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int fail(const char* what, int rc)
{
    std::fprintf(stderr, "gp_polish_files: %s failed (%d): %s\n", what, rc, gp_last_error());
    return 1;
}

int main(int argc, char** argv)
{
    gp_params prm;
    gp_default_params(&prm);
    std::vector<uint32_t> ks;
    uint32_t bsize = 1;
    int device = 0;
    double s_per_10kbp = -1, x_per_10kbp = 150;   // scripts/goldpolish:54-55,60: s 40 (PAF/SAM) or 100 (ntLink), x 150
    uint64_t max_bp_per_job = 256ull << 20;
    std::vector<const char*> pos;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](const char* opt) -> const char* {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "gp_polish_files: %s needs a value\n", opt);
                std::exit(1);
            }
            return argv[++i];
        };
        if (a == "-k") ks.push_back((uint32_t)std::strtoul(need("-k"), nullptr, 10));
        else if (a == "-b") bsize = (uint32_t)std::strtoul(need("-b"), nullptr, 10);
        else if (a == "-s") s_per_10kbp = std::strtod(need("-s"), nullptr);
        else if (a == "-x") x_per_10kbp = std::strtod(need("-x"), nullptr);
        else if (a == "-d") device = (int)std::strtol(need("-d"), nullptr, 10);
        else if (a == "-j") max_bp_per_job = std::strtoull(need("-j"), nullptr, 10) << 20;
        else pos.push_back(argv[i]);
    }
    if (pos.size() != 4 || bsize == 0) {
        std::fprintf(stderr, "usage: gp_polish_files [-k K]... [-b bsize] [-s S] [-x X] [-d device] [-j max_Mbp_per_job] "
                             "targets.fa reads.fq mappings.(paf|sam|tsv) out.fa\n");
        return 1;
    }
    if (!ks.empty()) {
        if (ks.size() > GP_MAX_K_VALUES) {
            std::fprintf(stderr, "gp_polish_files: at most %d k values\n", GP_MAX_K_VALUES);
            return 1;
        }
        prm.n_k = (uint32_t)ks.size();
        for (size_t i = 0; i < ks.size(); i++)
            prm.k[i] = ks[i];
    }
    const std::string map_path = pos[2];
    auto ends_with = [&](const char* suf) {
        const size_t n = std::strlen(suf);
        return map_path.size() >= n && map_path.compare(map_path.size() - n, n, suf) == 0;
    };
    if (s_per_10kbp < 0)
        s_per_10kbp = (ends_with(".paf") || ends_with(".sam")) ? 40 : 100; // scripts/goldpolish:227-249

    int rc;
    gp_seqindex *targets = nullptr, *reads = nullptr;
    gp_mappings* maps = nullptr;
    gp_ctx* ctx = nullptr;
    if ((rc = gp_seqindex_build(pos[0], &targets)) != GP_OK) return fail("indexing the targets", rc);
    if ((rc = gp_seqindex_build(pos[1], &reads)) != GP_OK) return fail("indexing the reads", rc);
    if ((rc = gp_mappings_load(pos[2], targets, 1, 30, x_per_10kbp, &maps)) != GP_OK) return fail("loading the mappings", rc);
    if ((rc = gp_ctx_create(device, &ctx)) != GP_OK) return fail("gp_ctx_create", rc);

    // batches in file order; a job takes whole batches up to max_bp_per_job contig bases
    const uint64_t n = gp_seqindex_count(targets);
    uint64_t done = 0, total_bp = 0;
    bool first_job = true;
    gp_counters sum{};
    while (done < n) {
        std::vector<const char*> ids;
        std::vector<uint32_t> batch_begin{ 0 };
        uint64_t bp = 0;
        uint64_t i = done;
        while (i < n) {
            uint64_t batch_bp = 0;
            const uint64_t batch_end = std::min<uint64_t>(n, (i / bsize + 1) * (uint64_t)bsize);
            for (uint64_t j = i; j < batch_end; j++) {
                const char* id; uint64_t st, len; double ph;
                if ((rc = gp_seqindex_entry(targets, j, &id, &st, &len, &ph)) != GP_OK) return fail("gp_seqindex_entry", rc);
                batch_bp += len;
            }
            if (!ids.empty() && bp + batch_bp > max_bp_per_job)
                break;
            for (uint64_t j = i; j < batch_end; j++) {
                const char* id; uint64_t st, len; double ph;
                gp_seqindex_entry(targets, j, &id, &st, &len, &ph);
                ids.push_back(id);
            }
            batch_begin.push_back((uint32_t)ids.size());
            bp += batch_bp;
            i = batch_end;
        }
        gp_job* job = nullptr;
        if ((rc = gp_job_create_from_files(ctx, &prm, targets, reads, maps, s_per_10kbp,
                                           (uint32_t)batch_begin.size() - 1, batch_begin.data(), ids.data(), &job)) != GP_OK)
            return fail("gp_job_create_from_files", rc);
        if ((rc = gp_job_polish(job, 1)) != GP_OK) return fail("gp_job_polish", rc);
        if ((rc = gp_job_write_fasta(job, nullptr, pos[3], first_job ? 0 : 1)) != GP_OK) return fail("gp_job_write_fasta", rc);
        gp_counters c;
        gp_job_get_counters(job, &c);
        sum.b_edits += c.b_edits; sum.c_gaps += c.c_gaps; sum.c_closed += c.c_closed;
        sum.c_no_start_goal += c.c_no_start_goal;
        gp_job_destroy(job);
        first_job = false;
        done = i;
        total_bp += bp;
    }
    // the two lines goldpolish-polish-batch greps out of Sealer's log when verbose (:111-129)
    std::fprintf(stderr, "No start/goal kmer: %llu\nGaps closed = %llu\n",
                 (unsigned long long)sum.c_no_start_goal, (unsigned long long)sum.c_closed);
    std::fprintf(stderr, "gp_polish_files: %llu records, %llu bp, %llu ntEdit edits, %llu of %llu gaps closed\n",
                 (unsigned long long)n, (unsigned long long)total_bp, (unsigned long long)sum.b_edits,
                 (unsigned long long)sum.c_closed, (unsigned long long)sum.c_gaps);
    gp_ctx_destroy(ctx);
    gp_mappings_destroy(maps);
    gp_seqindex_destroy(reads);
    gp_seqindex_destroy(targets);
    return 0;
}
