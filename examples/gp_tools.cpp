// gp_tools.cpp -- the three programs scripts/goldpolish-polish-batch expects on PATH for the path's
// device stages, as thin argv front ends over the C ABI (include/gpcore.h), so that the build can
// be dropped in under the reference's unmodified scripts.  One binary; what it
// does follows the name it is called by (the Makefile links the three names to it):
//
//   goldpolish-targeted-bfs <asm> <asm.index> <mappings> <reads> <reads.index> <x> <s> <threads> <k>...
//       the filter server of src/goldpolish_targeted_bfs.cpp:150-300: FIFOs in the working directory
//       (batch_name_input, batch_target_ids_input_ready; per batch <n>-target_ids_input and
//       <n>-bfs_ready), one batch name per line until "x"; per batch the target ids arrive on the
//       batch's FIFO, the filters are built on the GPU (gp_job_create_from_files +
//       gp_job_build_filters) and written as <n>-k<k>.bf, then "1" goes to <n>-bfs_ready.
//   ntedit-gr -f <in.fa> -r <bf> -b <prefix> [-d N -i N -m 1 -X x -Y y -j N -z N -a 0|1 -t N]
//       one ntEdit pass (subprojects/ntedit/ntedit.cpp: k and the number of hashes come from the
//       filter file, :2012-2022) over every sequence of in.fa, written to <prefix>_edited.fa.
//   sealer-gr -S <in.fa> -o <prefix> -L N -P N -B N [-G N -C N] -k K... --input-bloom=F... [-v --lower -j N]
//       Sealer's gap filling (subprojects/sealer/sealer.cpp:738-1115) with one filter per -k, written
//       to <prefix>_scaffold.fa; with -v the two lines goldpolish-polish-batch greps for go to stderr.
//
// Filters travel in btllib's file layout as gp_filter_save_btllib / gp_filter_load write and read it
// (UNPINNED: no btllib here); every tool of a chain must be this build's unless
// that layout has been checked against a btllib-written file.  Each call of ntedit-gr / sealer-gr is
// a process with a GPU context of its own: a compatibility path, not the fast one (that is
// gp_job_polish / gp_pipeline_* in the driver's process).  Plain C++17, no HIP in this file.
#include "../include/gpcore.h"

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

namespace {

[[noreturn]] void die(const std::string& tool, const std::string& what, int rc = 0)
{
    std::fprintf(stderr, "%s: %s%s%s\n", tool.c_str(), what.c_str(), rc ? ": " : "", rc ? gp_last_error() : "");
    std::exit(1);
}

struct Fasta {
    std::vector<std::string> headers; // "id[ comment]"
    std::vector<uint64_t> off{ 0 };
    std::string seq;
};

Fasta read_fasta(const std::string& tool, const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in)
        die(tool, "cannot open " + path);
    Fasta f;
    std::string line;
    bool open = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r')
            line.pop_back();
        if (!line.empty() && line[0] == '>') {
            if (open)
                f.off.push_back(f.seq.size());
            f.headers.push_back(line.substr(1));
            open = true;
        } else if (open) {
            f.seq += line;
        }
    }
    if (open)
        f.off.push_back(f.seq.size());
    return f;
}

// every sequence of `f` as one batch without reads
gp_job* job_of(const std::string& tool, gp_ctx* ctx, const gp_params& p, const Fasta& f)
{
    const uint32_t n = (uint32_t)f.headers.size();
    std::vector<uint32_t> bcb{ 0, n }, name_len(n), crb(n + 1, 0), thr(n, 4);
    for (uint32_t c = 0; c < n; c++)
        name_len[c] = (uint32_t)f.headers[c].size();
    gp_batch_set bs{};
    bs.n_batches = 1;
    bs.batch_contig_begin = bcb.data();
    bs.n_contigs = n;
    bs.contig_buf = reinterpret_cast<const uint8_t*>(f.seq.data());
    bs.contig_off = f.off.data();
    bs.contig_name_len = name_len.data();
    bs.contig_read_begin = crb.data();
    bs.contig_kmer_threshold = thr.data();
    uint64_t roff0 = 0;
    bs.n_reads = 0;
    bs.read_off = &roff0;
    gp_job* job = nullptr;
    int rc = gp_job_create(ctx, &p, &bs, &job);
    if (rc != GP_OK)
        die(tool, "gp_job_create", rc);
    return job;
}

void write_out(const std::string& tool, gp_job* job, const Fasta& f, const std::string& path)
{
    std::vector<const char*> names;
    for (const auto& h : f.headers)
        names.push_back(h.c_str());
    int rc = gp_job_write_fasta(job, names.data(), path.c_str(), 0);
    if (rc != GP_OK)
        die(tool, "writing " + path, rc);
}

// ------------------------------------------------------------------ ntedit-gr
int ntedit_main(int argc, char** argv)
{
    const std::string tool = "ntedit-gr";
    std::string in, bf, prefix;
    gp_params p;
    gp_default_params(&p);
    p.run_sealer = 0;
    auto val = [&](int& i, const char* a) -> std::string { // -x VALUE or -xVALUE
        if (a[2])
            return a + 2;
        if (i + 1 >= argc)
            die(tool, std::string("option ") + a + " needs a value");
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        if (a[0] != '-' || !a[1])
            die(tool, std::string("unexpected argument ") + a);
        switch (a[1]) {
        case 'f': in = val(i, a); break;
        case 'r': bf = val(i, a); break;
        case 'b': prefix = val(i, a); break;
        case 'd': p.max_deletions = (uint32_t)std::stoul(val(i, a)); break;
        case 'i': p.max_insertions = (uint32_t)std::stoul(val(i, a)); break;
        case 'm': p.mode = std::stoi(val(i, a)); break;
        case 'X': p.missing_ratio = std::stof(val(i, a)); break;
        case 'Y': p.edit_ratio = std::stof(val(i, a)); break;
        case 'j': p.jump = (uint32_t)std::stoul(val(i, a)); break;
        case 'z': p.min_contig_len = (uint32_t)std::stoul(val(i, a)); break;
        case 'a': p.soft_mask = std::stoi(val(i, a)); break;
        case 't': (void)val(i, a); break; // threads: the device has its own
        case 'v': break;
        default: die(tool, std::string("option ") + a + " is not supported by this build (ntEdit as GoldPolish calls it: "
                                                       "-f -r -b -d -i -m1 -X -Y -j -z -a -t)");
        }
    }
    if (in.empty() || bf.empty() || prefix.empty())
        die(tool, "usage: ntedit-gr -f <in.fa> -r <filter.bf> -b <prefix> [-d5 -i5 -m1 -X0.5 -Y0.5 -a1 -t N]");
    uint32_t k = 0, h = 0;
    uint64_t bytes = 0;
    int rc = gp_filter_load(bf.c_str(), &k, &h, &bytes, nullptr, 0);
    if (rc != GP_OK)
        die(tool, "reading " + bf, rc);
    std::vector<uint8_t> bits(bytes);
    if ((rc = gp_filter_load(bf.c_str(), &k, &h, &bytes, bits.data(), bits.size())) != GP_OK)
        die(tool, "reading " + bf, rc);
    p.n_k = 1;
    p.k[0] = k; // ntEdit takes k and the number of hashes from the filter (ntedit.cpp:2013,2022)
    p.hash_num = h;
    p.bf_bytes = bytes;
    const Fasta f = read_fasta(tool, in);
    gp_ctx* ctx = nullptr;
    if ((rc = gp_ctx_create(0, &ctx)) != GP_OK)
        die(tool, "gp_ctx_create", rc);
    gp_job* job = job_of(tool, ctx, p, f);
    if ((rc = gp_job_set_filters(job, bits.data())) != GP_OK || (rc = gp_job_ntedit_pass(job, 0, 1)) != GP_OK)
        die(tool, "the ntEdit pass", rc);
    write_out(tool, job, f, prefix + "_edited.fa");
    gp_job_destroy(job);
    gp_ctx_destroy(ctx);
    return 0;
}

// ------------------------------------------------------------------ sealer-gr
int sealer_main(int argc, char** argv)
{
    const std::string tool = "sealer-gr";
    std::string in, prefix;
    std::vector<uint32_t> ks;
    std::vector<std::string> blooms;
    gp_params p;
    gp_default_params(&p);
    bool verbose = false;
    uint32_t flank = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&](size_t short_len) -> std::string {
            if (a.size() > short_len)
                return a.substr(short_len);
            if (i + 1 >= argc)
                die(tool, "option " + a + " needs a value");
            return argv[++i];
        };
        if (a == "-v" || a == "--verbose") verbose = true;
        else if (a == "--lower") ;
        else if (a.rfind("--input-bloom=", 0) == 0) blooms.push_back(a.substr(14));
        else if (a.rfind("-S", 0) == 0) in = val(2);
        else if (a.rfind("-o", 0) == 0) prefix = val(2);
        else if (a.rfind("-L", 0) == 0) flank = (uint32_t)std::stoul(val(2));
        else if (a.rfind("-j", 0) == 0) (void)val(2);
        else if (a.rfind("-P", 0) == 0) p.sealer_max_paths = (uint32_t)std::stoul(val(2));
        else if (a.rfind("-B", 0) == 0) p.sealer_max_branches = (uint32_t)std::stoul(val(2));
        else if (a.rfind("-G", 0) == 0) p.sealer_max_gap = (uint32_t)std::stoul(val(2));
        else if (a.rfind("-C", 0) == 0) p.sealer_max_cost = (uint32_t)std::stoul(val(2));
        else if (a.rfind("-k", 0) == 0) ks.push_back((uint32_t)std::stoul(val(2)));
        else die(tool, "option " + a + " is not supported by this build (Sealer as GoldPolish calls it: -v -S -o -L -j -P -B "
                                        "--lower -k... --input-bloom=...)");
    }
    if (in.empty() || prefix.empty() || ks.empty() || ks.size() != blooms.size() || ks.size() > GP_MAX_K_VALUES)
        die(tool, "usage: sealer-gr -S <in.fa> -o <prefix> -L<k0> -P10 -B1000 --lower -k<K>... --input-bloom=<bf>... (one filter per -k)");
    if (flank && flank != ks[0])
        die(tool, "-L other than the first -k is not supported (GoldPolish passes -L$(firstword K), scripts/goldpolish-make:70)");
    p.n_k = (uint32_t)ks.size();
    std::vector<uint8_t> all;
    for (size_t i = 0; i < ks.size(); i++) {
        uint32_t k = 0, h = 0;
        uint64_t bytes = 0;
        int rc = gp_filter_load(blooms[i].c_str(), &k, &h, &bytes, nullptr, 0);
        if (rc != GP_OK)
            die(tool, "reading " + blooms[i], rc);
        if (k != ks[i])
            die(tool, blooms[i] + " was built for k = " + std::to_string(k) + ", not " + std::to_string(ks[i]));
        if (i && (bytes != p.bf_bytes || h != p.hash_num))
            die(tool, "the filters differ in size or number of hashes");
        p.k[i] = k;
        p.hash_num = h;
        p.bf_bytes = bytes;
        all.resize((i + 1) * bytes);
        if ((rc = gp_filter_load(blooms[i].c_str(), &k, &h, &bytes, all.data() + i * bytes, bytes)) != GP_OK)
            die(tool, "reading " + blooms[i], rc);
    }
    const Fasta f = read_fasta(tool, in);
    gp_ctx* ctx = nullptr;
    int rc;
    if ((rc = gp_ctx_create(0, &ctx)) != GP_OK)
        die(tool, "gp_ctx_create", rc);
    gp_job* job = job_of(tool, ctx, p, f);
    if ((rc = gp_job_set_filters(job, all.data())) != GP_OK || (rc = gp_job_reset(job, 1)) != GP_OK ||
        (rc = gp_job_sealer(job, 1)) != GP_OK)
        die(tool, "gap filling", rc);
    write_out(tool, job, f, prefix + "_scaffold.fa");
    gp_counters c;
    gp_job_get_counters(job, &c);
    if (verbose) // the lines scripts/goldpolish-polish-batch:111-129 looks for
        std::fprintf(stderr, "No start/goal kmer: %llu\nGaps closed = %llu\n", (unsigned long long)c.c_no_start_goal,
                     (unsigned long long)c.c_closed);
    gp_job_destroy(job);
    gp_ctx_destroy(ctx);
    return 0;
}

// ------------------------------------------------------ goldpolish-targeted-bfs
std::string read_fifo_line(const std::string& path)
{
    std::ifstream in(path); // blocks until a writer opens the FIFO
    std::string line;
    std::getline(in, line);
    return line;
}

void confirm_fifo(const std::string& path)
{
    std::ofstream out(path);
    out << "1\n";
}

int bfs_main(int argc, char** argv)
{
    const std::string tool = "goldpolish-targeted-bfs";
    if (argc < 10)
        die(tool, "Wrong args. (<asm> <asm.index> <mappings> <reads> <reads.index> <x> <s> <threads> <k>...)");
    // parent death ends the server (src/utils.cpp:62-70)
    (new std::thread([] {
        while (getppid() != 1)
            std::this_thread::sleep_for(std::chrono::seconds(1));
        std::exit(EXIT_FAILURE);
    }))->detach();
    const double x = std::stod(argv[6]), s = std::stod(argv[7]);
    gp_params p;
    gp_default_params(&p);
    p.n_k = 0;
    for (int i = 9; i < argc && p.n_k < GP_MAX_K_VALUES; i++)
        p.k[p.n_k++] = (uint32_t)std::stoul(argv[i]);
    int rc;
    gp_seqindex *targets = nullptr, *reads = nullptr;
    gp_mappings* maps = nullptr;
    gp_ctx* ctx = nullptr;
    if ((rc = gp_seqindex_load(argv[2], argv[1], &targets)) != GP_OK) die(tool, "loading the target index", rc);
    if ((rc = gp_seqindex_load(argv[5], argv[4], &reads)) != GP_OK) die(tool, "loading the read index", rc);
    if ((rc = gp_mappings_load(argv[3], targets, 1, 30, x, &maps)) != GP_OK) die(tool, "loading the mappings", rc);
    if ((rc = gp_ctx_create(0, &ctx)) != GP_OK) die(tool, "gp_ctx_create", rc);
    const std::string name_pipe = "batch_name_input", ready_pipe = "batch_target_ids_input_ready";
    if (mkfifo(name_pipe.c_str(), S_IRUSR | S_IWUSR) != 0 || mkfifo(ready_pipe.c_str(), S_IRUSR | S_IWUSR) != 0)
        die(tool, std::string("mkfifo failed: ") + std::strerror(errno));
    std::fprintf(stderr, "%s: Accepting batch names at %s\n", tool.c_str(), name_pipe.c_str());
    for (;;) {
        const std::string batch = read_fifo_line(name_pipe);
        if (batch.empty() || batch == "x")
            break;
        const std::string ids_pipe = batch + "-target_ids_input", done_pipe = batch + "-bfs_ready";
        if (mkfifo(ids_pipe.c_str(), S_IRUSR | S_IWUSR) != 0 || mkfifo(done_pipe.c_str(), S_IRUSR | S_IWUSR) != 0)
            die(tool, std::string("mkfifo failed: ") + std::strerror(errno));
        confirm_fifo(ready_pipe);
        std::vector<std::string> ids;
        {
            std::ifstream in(ids_pipe);
            std::string id;
            while (in >> id && id != "x")
                ids.push_back(id);
        }
        std::vector<const char*> idp;
        for (const auto& i : ids)
            idp.push_back(i.c_str());
        const uint32_t begin[2] = { 0, (uint32_t)ids.size() };
        gp_job* job = nullptr;
        if ((rc = gp_job_create_from_files(ctx, &p, targets, reads, maps, s, 1, begin, idp.data(), &job)) != GP_OK)
            die(tool, "selecting the reads of batch " + batch, rc);
        if ((rc = gp_job_build_filters(job, 1)) != GP_OK)
            die(tool, "building the filters of batch " + batch, rc);
        std::vector<uint8_t> bits(p.bf_bytes);
        for (uint32_t ki = 0; ki < p.n_k; ki++) {
            if ((rc = gp_job_get_filter(job, 0, ki, bits.data())) != GP_OK)
                die(tool, "gp_job_get_filter", rc);
            const std::string path = batch + "-k" + std::to_string(p.k[ki]) + ".bf";
            if ((rc = gp_filter_save_btllib(path.c_str(), p.k[ki], p.hash_num, p.bf_bytes, bits.data())) != GP_OK)
                die(tool, "writing " + path, rc);
        }
        gp_job_destroy(job);
        confirm_fifo(done_pipe);
        std::remove(ids_pipe.c_str());
        std::remove(done_pipe.c_str());
    }
    std::remove(name_pipe.c_str());
    std::remove(ready_pipe.c_str());
    std::fprintf(stderr, "%s: Targeted BF builder done!\n", tool.c_str());
    gp_ctx_destroy(ctx);
    gp_mappings_destroy(maps);
    gp_seqindex_destroy(reads);
    gp_seqindex_destroy(targets);
    return 0;
}

} // namespace

int main(int argc, char** argv)
{
    std::string me = argv[0];
    const size_t slash = me.rfind('/');
    if (slash != std::string::npos)
        me = me.substr(slash + 1);
    if (me == "ntedit-gr")
        return ntedit_main(argc, argv);
    if (me == "sealer-gr")
        return sealer_main(argc, argv);
    if (me == "goldpolish-targeted-bfs")
        return bfs_main(argc, argv);
    if (argc > 1) { // gp_tools <tool> args...
        const std::string t = argv[1];
        argv[1] = argv[0];
        if (t == "ntedit-gr") return ntedit_main(argc - 1, argv + 1);
        if (t == "sealer-gr") return sealer_main(argc - 1, argv + 1);
        if (t == "goldpolish-targeted-bfs") return bfs_main(argc - 1, argv + 1);
    }
    std::fprintf(stderr, "gp_tools: call me as ntedit-gr, sealer-gr or goldpolish-targeted-bfs (or: gp_tools <tool> args...)\n");
    return 1;
}
