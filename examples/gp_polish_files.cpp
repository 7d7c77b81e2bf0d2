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
// Several GPUs of one node: `-d 0,1,...` starts one host thread with a gp_ctx of its own per
// device; the batches are dealt to them by gp_partition_batches (longest first, by
// contig bases + mapped read bases, onto the least loaded), every thread polishes its batches in
// file order into a part file of its own, and the part files are merged back into input order at
// the end -- what scripts/goldpolish-reaper:51-73 does with the batches of one node.
//
// usage: gp_polish_files [-k K]... [-b bsize] [-s reads_per_10kbp] [-x mx_per_10kbp] [-d device[,device...]]
//                        [-j max_Mbp_per_job] targets.fa reads.fq mappings.(paf|sam|tsv) out.fa
#include "../include/gpcore.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static int fail(const char* what, int rc)
{
    std::fprintf(stderr, "gp_polish_files: %s failed (%d): %s\n", what, rc, gp_last_error());
    return 1;
}

// one device's share: the batches listed in `mine` (ascending), polished job by job into `part`
struct Share {
    int device = 0;
    std::vector<uint64_t> mine;      // batch indices
    std::string part;                // its FASTA: records in ascending input order
    std::vector<uint64_t> records;   // input index of every record written (dropped ones are absent)
    gp_counters sum{};
    uint64_t bp = 0;
    int rc = GP_OK;
    std::string err;
};

int main(int argc, char** argv)
{
    gp_params prm;
    gp_default_params(&prm);
    std::vector<uint32_t> ks;
    uint32_t bsize = 1;
    std::vector<int> devices;
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
        else if (a == "-d") {
            for (const char* q = need("-d"); *q;) {
                char* end;
                devices.push_back((int)std::strtol(q, &end, 10));
                if (end == q) {
                    std::fprintf(stderr, "gp_polish_files: -d takes device numbers separated by commas\n");
                    return 1;
                }
                q = *end == ',' ? end + 1 : end;
            }
        }
        else if (a == "-j") max_bp_per_job = std::strtoull(need("-j"), nullptr, 10) << 20;
        else pos.push_back(argv[i]);
    }
    if (devices.empty())
        devices.push_back(0);
    if (pos.size() != 4 || bsize == 0) {
        std::fprintf(stderr, "usage: gp_polish_files [-k K]... [-b bsize] [-s S] [-x X] [-d device[,device...]] "
                             "[-j max_Mbp_per_job] targets.fa reads.fq mappings.(paf|sam|tsv) out.fa\n");
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
    const std::string map_path = pos[2], out_path = pos[3];
    auto ends_with = [&](const char* suf) {
        const size_t n = std::strlen(suf);
        return map_path.size() >= n && map_path.compare(map_path.size() - n, n, suf) == 0;
    };
    if (s_per_10kbp < 0)
        s_per_10kbp = (ends_with(".paf") || ends_with(".sam")) ? 40 : 100; // scripts/goldpolish:227-249

    int rc;
    gp_seqindex *targets = nullptr, *reads = nullptr;
    gp_mappings* maps = nullptr;
    if ((rc = gp_seqindex_build(pos[0], &targets)) != GP_OK) return fail("indexing the targets", rc);
    if ((rc = gp_seqindex_build(pos[1], &reads)) != GP_OK) return fail("indexing the reads", rc);
    if ((rc = gp_mappings_load(pos[2], targets, 1, 30, x_per_10kbp, &maps)) != GP_OK) return fail("loading the mappings", rc);

    // ---- batches in file order (record.num % bsize == bsize - 1 closes one) and what each costs
    const uint64_t n = gp_seqindex_count(targets);
    const uint64_t n_batches = (n + bsize - 1) / bsize;
    std::vector<const char*> id_of(n);
    std::vector<uint64_t> len_of(n), cost(n_batches, 0);
    for (uint64_t j = 0; j < n; j++) {
        uint64_t st; double ph;
        if ((rc = gp_seqindex_entry(targets, j, &id_of[j], &st, &len_of[j], &ph)) != GP_OK) return fail("gp_seqindex_entry", rc);
        cost[j / bsize] += len_of[j];
        if (devices.size() > 1) { // contig bases + mapped read bases
            uint64_t n_ids = 0;
            const char* const* rids = nullptr;
            if ((rc = gp_mappings_get(maps, id_of[j], &n_ids, &rids)) != GP_OK) return fail("gp_mappings_get", rc);
            for (uint64_t r = 0; r < n_ids; r++) {
                uint64_t rs, rl; double rp;
                if (gp_seqindex_find(reads, rids[r], &rs, &rl, &rp) == GP_OK)
                    cost[j / bsize] += rl;
            }
        }
    }
    std::vector<uint32_t> rank_of(n_batches ? n_batches : 1, 0);
    if (n_batches >= (1ull << 32)) {
        std::fprintf(stderr, "gp_polish_files: too many batches\n");
        return 1;
    }
    if ((rc = gp_partition_batches((uint32_t)n_batches, cost.data(), (uint32_t)devices.size(), rank_of.data(), nullptr)) != GP_OK)
        return fail("gp_partition_batches", rc);
    std::vector<Share> shares(devices.size());
    for (size_t d = 0; d < devices.size(); d++) {
        shares[d].device = devices[d];
        shares[d].part = devices.size() == 1 ? out_path : out_path + ".part" + std::to_string(d);
    }
    for (uint64_t b = 0; b < n_batches; b++)
        shares[rank_of[b]].mine.push_back(b);

    // ---- one host thread per device.  The index and the mappings are shared: job creation (which
    // reads through them) takes turns, the polishing runs side by side.
    std::mutex create_mu;
    auto work = [&](Share& sh) {
        auto failed = [&](const char* what, int code) {
            sh.rc = code;
            sh.err = std::string(what) + " failed: " + gp_last_error();
        };
        gp_ctx* ctx = nullptr;
        int r;
        if ((r = gp_ctx_create(sh.device, &ctx)) != GP_OK)
            return failed("gp_ctx_create", r);
        bool first_job = true;
        size_t at = 0;
        while (at < sh.mine.size() && sh.rc == GP_OK) {
            // a job takes whole batches up to max_bp_per_job contig bases
            std::vector<const char*> ids;
            std::vector<uint64_t> rec;
            std::vector<uint32_t> batch_begin{ 0 };
            uint64_t bp = 0;
            while (at < sh.mine.size()) {
                const uint64_t b = sh.mine[at], j0 = b * bsize, j1 = std::min<uint64_t>(n, j0 + bsize);
                uint64_t batch_bp = 0;
                for (uint64_t j = j0; j < j1; j++)
                    batch_bp += len_of[j];
                if (!ids.empty() && bp + batch_bp > max_bp_per_job)
                    break;
                for (uint64_t j = j0; j < j1; j++) {
                    ids.push_back(id_of[j]);
                    rec.push_back(j);
                }
                batch_begin.push_back((uint32_t)ids.size());
                bp += batch_bp;
                at++;
            }
            gp_job* job = nullptr;
            {
                std::lock_guard<std::mutex> g(create_mu);
                r = gp_job_create_from_files(ctx, &prm, targets, reads, maps, s_per_10kbp, (uint32_t)batch_begin.size() - 1,
                                             batch_begin.data(), ids.data(), &job);
            }
            if (r != GP_OK) {
                failed("gp_job_create_from_files", r);
                break;
            }
            std::vector<uint64_t> lens(ids.size());
            std::vector<uint8_t> dropped(ids.size());
            uint64_t total = 0;
            if ((r = gp_job_polish(job, 1)) != GP_OK) failed("gp_job_polish", r);
            else if ((r = gp_job_get_lengths(job, lens.data(), dropped.data(), &total)) != GP_OK) failed("gp_job_get_lengths", r);
            else if ((r = gp_job_write_fasta(job, nullptr, sh.part.c_str(), first_job ? 0 : 1)) != GP_OK) failed("gp_job_write_fasta", r);
            if (sh.rc == GP_OK) {
                for (size_t i = 0; i < rec.size(); i++)
                    if (!dropped[i])
                        sh.records.push_back(rec[i]);
                gp_counters c;
                gp_job_get_counters(job, &c);
                sh.sum.b_edits += c.b_edits; sh.sum.c_gaps += c.c_gaps; sh.sum.c_closed += c.c_closed;
                sh.sum.c_no_start_goal += c.c_no_start_goal;
                sh.bp += bp;
            }
            gp_job_destroy(job);
            first_job = false;
        }
        if (first_job && sh.rc == GP_OK) // no batch at all: still an (empty) part
            std::ofstream(sh.part, std::ios::binary | std::ios::trunc);
        gp_ctx_destroy(ctx);
    };
    std::vector<std::thread> threads;
    for (size_t d = 1; d < shares.size(); d++)
        threads.emplace_back(work, std::ref(shares[d]));
    work(shares[0]);
    for (auto& t : threads)
        t.join();
    gp_counters sum{};
    uint64_t total_bp = 0;
    for (const Share& sh : shares) {
        if (sh.rc != GP_OK) {
            std::fprintf(stderr, "gp_polish_files: device %d: %s (%d)\n", sh.device, sh.err.c_str(), sh.rc);
            return 1;
        }
        sum.b_edits += sh.sum.b_edits; sum.c_gaps += sh.sum.c_gaps; sum.c_closed += sh.sum.c_closed;
        sum.c_no_start_goal += sh.sum.c_no_start_goal;
        total_bp += sh.bp;
    }

    // ---- the parts back into input order: every part is ascending, so this is a merge
    if (shares.size() > 1) {
        std::vector<std::ifstream> in(shares.size());
        std::vector<size_t> next(shares.size(), 0);
        for (size_t d = 0; d < shares.size(); d++)
            in[d].open(shares[d].part, std::ios::binary);
        std::ofstream out(out_path, std::ios::binary | std::ios::trunc);
        std::string header, seq;
        for (;;) {
            size_t best = shares.size();
            for (size_t d = 0; d < shares.size(); d++)
                if (next[d] < shares[d].records.size() &&
                    (best == shares.size() || shares[d].records[next[d]] < shares[best].records[next[best]]))
                    best = d;
            if (best == shares.size())
                break;
            if (!std::getline(in[best], header) || !std::getline(in[best], seq) || header.empty() || header[0] != '>') {
                std::fprintf(stderr, "gp_polish_files: %s is shorter than the records written to it\n", shares[best].part.c_str());
                return 1;
            }
            out << header << '\n' << seq << '\n';
            next[best]++;
        }
        out.flush();
        if (!out) {
            std::fprintf(stderr, "gp_polish_files: writing %s failed\n", out_path.c_str());
            return 1;
        }
        for (const Share& sh : shares)
            std::remove(sh.part.c_str());
    }
    // the two lines goldpolish-polish-batch greps out of Sealer's log when verbose (:111-129)
    std::fprintf(stderr, "No start/goal kmer: %llu\nGaps closed = %llu\n",
                 (unsigned long long)sum.c_no_start_goal, (unsigned long long)sum.c_closed);
    std::fprintf(stderr, "gp_polish_files: %llu records, %llu bp on %zu device(s), %llu ntEdit edits, %llu of %llu gaps closed\n",
                 (unsigned long long)n, (unsigned long long)total_bp, shares.size(), (unsigned long long)sum.b_edits,
                 (unsigned long long)sum.c_closed, (unsigned long long)sum.c_gaps);
    gp_mappings_destroy(maps);
    gp_seqindex_destroy(reads);
    gp_seqindex_destroy(targets);
    return 0;
}
