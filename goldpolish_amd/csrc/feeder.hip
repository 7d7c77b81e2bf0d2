// feeder.hip -- the step in front of stage A, on the host: which reads reach the filter
// build, and in which order.  Plain C++ (no kernels here); it lives in
// libgpcore so that a caller holding only file names can get a device-resident job.
//
// What it replaces in the reference:
//   * SeqIndex            /root/reference/src/seqindex.cpp:12-137, seqindex.hpp:59-102
//                         (`goldpolish-index`, src/goldpolish_index.cpp)
//   * AllMappings         /root/reference/src/mappings.cpp:15-345
//   * serve_batch's read selection
//                         /root/reference/src/goldpolish_targeted_bfs.cpp:86-131
// The behaviour that is easy to get wrong is kept on purpose: the FASTQ id stops at the first
// space and then at the first tab, the FASTA id only at the first space; the average phred
// leaves the last quality character out (calc_phred_avg(line, 0, size-1)); a saved index holds
// phred with six significant digits; a short SAM/PAF line inherits the previous line's target;
// mappings are de-duplicated per target in first-seen order and dropped when the target is
// not indexed; only the 3-column ntLink table is filtered by common minimizers.
#include "gp_internal.h"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

struct gp_seqindex {
    struct Entry {
        std::string id;
        uint64_t start, len;
        double phred;
    };
    std::string seqs_path;
    std::vector<Entry> entries;                       // file order
    std::unordered_map<std::string, uint32_t> by_id;  // first record of an id wins (emplace)
    int fd = -1;
    ~gp_seqindex()
    {
        if (fd >= 0)
            close(fd);
    }
    const Entry* find(const std::string& id) const
    {
        auto it = by_id.find(id);
        return it == by_id.end() ? nullptr : &entries[it->second];
    }
    void add(std::string id, uint64_t start, uint64_t len, double phred)
    {
        if (by_id.emplace(id, (uint32_t)entries.size()).second)
            entries.push_back({ std::move(id), start, len, phred });
    }
};

struct gp_mappings {
    std::unordered_map<std::string, std::vector<std::string>> by_target;
    std::vector<const char*> scratch; // what gp_mappings_get hands out
};

namespace gp {
namespace {

std::string until(const std::string& s, char sep)
{
    const size_t p = s.find(sep);
    return p == std::string::npos ? s : s.substr(0, p);
}

// btllib::calc_phred_avg(qual, 0, len) as the index calls it: len == 0 means "all of it"
double phred_avg(const std::string& qual, size_t len)
{
    if (len == 0)
        len = qual.size();
    if (len == 0)
        return -33.0;
    uint64_t sum = 0;
    for (size_t i = 0; i < len; i++)
        sum += (unsigned char)qual[i];
    return (double)sum / (double)len - 33.0;
}

bool ends_with(const std::string& s, const char* suffix)
{
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

int read_exact(const gp_seqindex* ix, uint64_t start, uint64_t len, uint8_t* out)
{
    uint64_t done = 0;
    while (done < len) {
        const ssize_t r = pread(ix->fd, out + done, len - done, (off_t)(start + done));
        if (r <= 0) {
            set_error("read of " + ix->seqs_path + " failed: " + (r < 0 ? std::strerror(errno) : "short file"));
            return GP_ERR_INVALID_ARG;
        }
        done += (uint64_t)r;
    }
    return GP_OK;
}

int open_seqs(gp_seqindex* ix)
{
    ix->fd = open(ix->seqs_path.c_str(), O_RDONLY);
    if (ix->fd < 0) {
        set_error("cannot open " + ix->seqs_path + ": " + std::strerror(errno));
        return GP_ERR_INVALID_ARG;
    }
    (void)posix_fadvise(ix->fd, 0, 0, POSIX_FADV_RANDOM); // fetches are random (seqindex.hpp:76)
    return GP_OK;
}

// one target's mappings while loading
struct TargetMappings {
    std::vector<std::string> ids;
    std::vector<unsigned> mx;
    std::unordered_set<std::string> seen;
};
using Loading = std::unordered_map<std::string, TargetMappings>;

void load_mapping(Loading& all, const std::string& mapped, const std::string& target,
                  const gp_seqindex* targets, unsigned mx)
{
    if (!targets->find(target))
        return;
    TargetMappings& t = all[target];
    if (t.seen.insert(mapped).second) {
        t.ids.push_back(mapped);
        t.mx.push_back(mx);
    }
}

// SAM: QNAME is column 1, RNAME column 3; PAF: query column 1, target column 6.  Lines that
// start with '@' are skipped in both (mappings.cpp:152,207).
int load_columns(const std::string& path, const gp_seqindex* targets, int target_col, Loading& all)
{
    std::ifstream in(path);
    if (!in) {
        set_error("cannot open " + path);
        return GP_ERR_INVALID_ARG;
    }
    std::string line, token, mapped, target; // mapped/target outlive the line, as in the reference
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == '@')
            continue;
        std::stringstream ss(line);
        int col = 1;
        while (ss >> token) {
            if (col == 1)
                mapped = token;
            else if (col == target_col)
                target = token;
            col++;
        }
        load_mapping(all, mapped, target, targets, 0);
    }
    return GP_OK;
}

unsigned count_at_least(const std::vector<unsigned>& mx, unsigned thr)
{
    unsigned n = 0;
    for (unsigned v : mx)
        n += v >= thr;
    return n;
}

// the common-minimizer threshold of one target (mappings.cpp:231-320)
int filter_target(TargetMappings& t, uint64_t target_len, double max_per_10kbp, unsigned mx_min,
                  unsigned mx_max)
{
    if (t.ids.empty())
        return GP_OK;
    const int max_mapped = (int)std::ceil((double)target_len * max_per_10kbp / 10000.0);
    if (max_mapped <= 0) {
        set_error("mappings filter: max_mapped_seqs <= 0");
        return GP_ERR_INVALID_ARG;
    }
    int lo = (int)mx_min, hi = (int)mx_max;
    const int lo_n = (int)t.ids.size();
    const int hi_n = (int)count_at_least(t.mx, (unsigned)hi);
    int thr;
    if (lo_n <= max_mapped) {
        thr = lo;
    } else if (hi_n > max_mapped) {
        thr = hi;
    } else {
        while (hi - lo > 1) {
            const int mid = (hi + lo) / 2;
            if ((int)count_at_least(t.mx, (unsigned)mid) > max_mapped)
                lo = mid;
            else
                hi = mid;
        }
        thr = hi;
    }
    std::vector<std::string> kept;
    for (size_t i = 0; i < t.ids.size(); i++)
        if ((int)t.mx[i] >= thr)
            kept.push_back(std::move(t.ids[i]));
    t.ids.swap(kept);
    return GP_OK;
}

} // namespace
} // namespace gp

using namespace gp;

extern "C" {

/* SeqIndex(seqs_filepath), src/seqindex.cpp:12-66 */
int gp_seqindex_build(const char* seqs_path, gp_seqindex** out)
try {
    if (!seqs_path || !out)
        return GP_ERR_INVALID_ARG;
    std::ifstream in(seqs_path);
    if (!in) {
        set_error(std::string("cannot open ") + seqs_path);
        return GP_ERR_INVALID_ARG;
    }
    auto* ix = new gp_seqindex;
    ix->seqs_path = seqs_path;
    const bool fastq = in.peek() == '@';
    std::string line, id;
    uint64_t i = 0, byte = 0, id_end = 0, seq_start = 0, seq_len = 0;
    while (std::getline(in, line)) {
        const uint64_t end = byte + line.size();
        if (fastq) {
            switch (i % 4) {
            case 0:
                id_end = end;
                id = until(until(line, ' ').substr(line.empty() ? 0 : 1), '\t');
                break;
            case 1:
                seq_start = id_end + 1;
                seq_len = end - id_end - 1;
                break;
            case 3:
                ix->add(id, seq_start, seq_len, phred_avg(line, line.empty() ? 0 : line.size() - 1));
                break;
            default:
                break;
            }
        } else {
            if (i % 2 == 0) {
                id_end = end;
                id = until(line, ' ').substr(line.empty() ? 0 : 1);
            } else {
                ix->add(id, id_end + 1, end - id_end - 1, 0.0);
            }
        }
        byte = end + 1;
        i++;
    }
    int rc = open_seqs(ix);
    if (rc != GP_OK) {
        delete ix;
        return rc;
    }
    *out = ix;
    return GP_OK;
} GP_CATCH("gp_seqindex_build")

/* SeqIndex(index_filepath, seqs_filepath), src/seqindex.cpp:85-125 */
int gp_seqindex_load(const char* index_path, const char* seqs_path, gp_seqindex** out)
try {
    if (!index_path || !seqs_path || !out)
        return GP_ERR_INVALID_ARG;
    std::ifstream in(index_path);
    if (!in) {
        set_error(std::string("cannot open ") + index_path);
        return GP_ERR_INVALID_ARG;
    }
    auto* ix = new gp_seqindex;
    ix->seqs_path = seqs_path;
    std::string token, id;
    uint64_t start = 0, len = 0, i = 0;
    try {
        while (in >> token) {
            switch (i % 4) {
            case 0: id = token; break;
            case 1: start = std::stoul(token); break;
            case 2: len = std::stoul(token); break;
            default: ix->add(id, start, len, std::stod(token)); break;
            }
            i++;
        }
    } catch (const std::exception&) {
        set_error(std::string("malformed index ") + index_path);
        delete ix;
        return GP_ERR_INVALID_ARG;
    }
    int rc = open_seqs(ix);
    if (rc != GP_OK) {
        delete ix;
        return rc;
    }
    *out = ix;
    return GP_OK;
} GP_CATCH("gp_seqindex_load")

/* SeqIndex::save, src/seqindex.cpp:68-83: id, start, length, average phred as operator<< prints a
 * double (six significant digits).  Lines come in file order here; the reference's order is
 * that of its hash map and the loader does not care. */
int gp_seqindex_save(const gp_seqindex* ix, const char* index_path)
try {
    if (!ix || !index_path)
        return GP_ERR_INVALID_ARG;
    FILE* f = std::fopen(index_path, "w");
    if (!f) {
        set_error(std::string("cannot write ") + index_path);
        return GP_ERR_INVALID_ARG;
    }
    for (const auto& e : ix->entries)
        std::fprintf(f, "%s\t%llu\t%llu\t%g\n", e.id.c_str(), (unsigned long long)e.start,
                     (unsigned long long)e.len, e.phred);
    std::fclose(f);
    return GP_OK;
} GP_CATCH("gp_seqindex_save")

uint64_t gp_seqindex_count(const gp_seqindex* ix) { return ix ? ix->entries.size() : 0; }

int gp_seqindex_entry(const gp_seqindex* ix, uint64_t i, const char** id, uint64_t* start,
                      uint64_t* len, double* phred)
try {
    if (!ix || i >= ix->entries.size())
        return GP_ERR_INVALID_ARG;
    const auto& e = ix->entries[i];
    if (id)
        *id = e.id.c_str();
    if (start)
        *start = e.start;
    if (len)
        *len = e.len;
    if (phred)
        *phred = e.phred;
    return GP_OK;
} GP_CATCH("gp_seqindex_entry")

int gp_seqindex_find(const gp_seqindex* ix, const char* id, uint64_t* start, uint64_t* len,
                     double* phred)
try {
    if (!ix || !id)
        return GP_ERR_INVALID_ARG;
    const auto* e = ix->find(id);
    if (!e)
        return GP_ERR_NOT_FOUND;
    if (start)
        *start = e->start;
    if (len)
        *len = e->len;
    if (phred)
        *phred = e->phred;
    return GP_OK;
} GP_CATCH("gp_seqindex_find")

/* SeqIndex::get_seq, src/seqindex.hpp:59-102 (no 20 MiB limit here: the caller sizes buf) */
int gp_seqindex_read(const gp_seqindex* ix, const char* id, uint8_t* buf, uint64_t buf_bytes,
                     uint64_t* len)
try {
    if (!ix || !id)
        return GP_ERR_INVALID_ARG;
    const auto* e = ix->find(id);
    if (!e) {
        set_error(std::string("sequence not in index: ") + id);
        return GP_ERR_NOT_FOUND;
    }
    if (len)
        *len = e->len;
    if (!buf)
        return GP_OK;
    if (buf_bytes < e->len) {
        set_error("gp_seqindex_read: buffer too small");
        return GP_ERR_INVALID_ARG;
    }
    return read_exact(ix, e->start, e->len, buf);
} GP_CATCH("gp_seqindex_read")

void gp_seqindex_destroy(gp_seqindex* ix) { delete ix; }

/* AllMappings::AllMappings, src/mappings.cpp:15-36 */
int gp_mappings_load(const char* path, const gp_seqindex* targets, uint32_t mx_min, uint32_t mx_max,
                     double mx_max_mapped_per_10kbp, gp_mappings** out)
try {
    if (!path || !targets || !out)
        return GP_ERR_INVALID_ARG;
    const std::string p = path;
    Loading all;
    int rc = GP_OK;
    if (ends_with(p, ".bam")) {
        set_error("BAM needs samtools (the reference pipes it through btllib::DataSource): convert to .sam");
        return GP_ERR_UNSUPPORTED;
    } else if (ends_with(p, ".sam")) {
        rc = load_columns(p, targets, 3, all);
    } else if (ends_with(p, ".paf")) {
        rc = load_columns(p, targets, 6, all);
    } else {
        // ntLink: "read target minimizers" (scripts/goldpolish-make:56-57), mappings.cpp:78-113
        std::ifstream in(p);
        if (!in) {
            set_error("cannot open " + p);
            return GP_ERR_INVALID_ARG;
        }
        if (mx_max_mapped_per_10kbp <= 0 || mx_min >= mx_max) {
            set_error("mappings filter: needs max_mapped_per_10kbp > 0 and mx_min < mx_max");
            return GP_ERR_INVALID_ARG;
        }
        std::string token, mapped, target;
        uint64_t i = 0;
        try {
            while (in >> token) {
                if (i % 3 == 0) {
                    mapped = token;
                } else if (i % 3 == 1) {
                    target = token;
                } else {
                    const unsigned long mx = std::stoul(token);
                    if (mx >= mx_min)
                        load_mapping(all, mapped, target, targets, (unsigned)mx);
                }
                i++;
            }
        } catch (const std::exception&) {
            set_error("malformed ntLink mapping table " + p);
            return GP_ERR_INVALID_ARG;
        }
        for (auto& kv : all) {
            const auto* e = targets->find(kv.first);
            if (!e)
                continue;
            if ((rc = filter_target(kv.second, e->len, mx_max_mapped_per_10kbp, mx_min, mx_max)) != GP_OK)
                return rc;
        }
    }
    if (rc != GP_OK)
        return rc;
    auto* m = new gp_mappings;
    for (auto& kv : all)
        m->by_target.emplace(kv.first, std::move(kv.second.ids));
    *out = m;
    return GP_OK;
} GP_CATCH("gp_mappings_load")

/* AllMappings::get_mappings, src/mappings.cpp:337-345.  The pointers stay valid until the next
 * gp_mappings_get on this object. */
int gp_mappings_get(gp_mappings* m, const char* target_id, uint64_t* n, const char* const** ids)
try {
    if (!m || !target_id || !n)
        return GP_ERR_INVALID_ARG;
    m->scratch.clear();
    auto it = m->by_target.find(target_id);
    if (it != m->by_target.end())
        for (const auto& s : it->second)
            m->scratch.push_back(s.c_str());
    *n = m->scratch.size();
    if (ids)
        *ids = m->scratch.data();
    return GP_OK;
} GP_CATCH("gp_mappings_get")

void gp_mappings_destroy(gp_mappings* m) { delete m; }

/* bfs[i]->save("<batch>-k<k>.bf") / KmerBloomFilter(path), src/goldpolish_targeted_bfs.cpp:138-140 and
 * subprojects/ntedit/ntedit.cpp:2013-2022 -- in this build's OWN container, not btllib's (that
 * header cannot be pinned here): the line "GPBF1 <k> <hash_num> <bytes>\n", then the
 * filter bytes (bit n -> byte n/8, mask 1 << n%8, the order btllib uses in memory). */
// header + payload into a new file; every write checked, a file that could not be written whole is removed
static int write_filter_file(const char* path, const std::string& header, uint64_t bytes, const uint8_t* data)
{
    FILE* f = std::fopen(path, "wb");
    if (!f) {
        set_error(std::string("cannot open ") + path + " for writing: " + std::strerror(errno));
        return GP_ERR_IO;
    }
    bool ok = std::fwrite(header.data(), 1, header.size(), f) == header.size() &&
              std::fwrite(data, 1, bytes, f) == bytes && std::fflush(f) == 0;
    const int e = errno;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) {
        (void)std::remove(path);
        set_error(std::string("writing ") + path + " failed: " + std::strerror(e));
        return GP_ERR_IO;
    }
    return GP_OK;
}

int gp_filter_save(const char* path, uint32_t k, uint32_t hash_num, uint64_t bytes, const uint8_t* data)
try {
    if (!path || !data)
        return GP_ERR_INVALID_ARG;
    return write_filter_file(path, "GPBF1 " + std::to_string(k) + " " + std::to_string(hash_num) + " " +
                                       std::to_string(bytes) + "\n", bytes, data);
} GP_CATCH("gp_filter_save")

/* The same filter under the header layout btllib publishes for KmerBloomFilter::save (SURVEY.md
 * 8a-A7): "[BTLKmerBloomFilter_v2]", the keys bytes / hash_fn / hash_num / k, "[HeaderEnd]", then
 * the raw bytes (bit n of the filter = byte n/8, mask 1 << n%8, which is how gp_job_get_filter
 * returns them).  UNPINNED like the reader below: written from the published layout, no btllib
 * here to read it back. */
int gp_filter_save_btllib(const char* path, uint32_t k, uint32_t hash_num, uint64_t bytes, const uint8_t* data)
try {
    if (!path || !data)
        return GP_ERR_INVALID_ARG;
    std::string h = "[BTLKmerBloomFilter_v2]\n";
    h += "bytes = " + std::to_string(bytes) + "\n";
    h += "hash_fn = \"ntHash_v2\"\n";
    h += "hash_num = " + std::to_string(hash_num) + "\n";
    h += "k = " + std::to_string(k) + "\n";
    h += "[HeaderEnd]\n";
    return write_filter_file(path, h, bytes, data);
} GP_CATCH("gp_filter_save_btllib")

/* Reader for the header SURVEY.md 8a-A7 describes for btllib's own files: a first line
 * "[BTLKmerBloomFilter_v<n>]", then "key = value" lines in any order (bytes, hash_num, k are needed,
 * hash_fn and anything else is skipped), then "[HeaderEnd]", then the filter bytes. UNPINNED: no
 * btllib source or btllib-written file exists in this image, so this follows the
 * published layout only and is exercised with hand-written headers; the payload must be exactly
 * `bytes` long or the file is refused. Leaves f at the first payload byte. */
static bool read_btl_header(FILE* f, unsigned* k, unsigned* hash_num, unsigned long long* bytes)
{
    char line[256];
    if (!std::fgets(line, sizeof line, f) || std::strncmp(line, "[BTLKmerBloomFilter_v", 21) != 0)
        return false;
    bool have_k = false, have_h = false, have_b = false, ended = false;
    while (std::fgets(line, sizeof line, f)) {
        if (std::strncmp(line, "[HeaderEnd]", 11) == 0) {
            ended = true;
            break;
        }
        char key[64];
        unsigned long long v = 0;
        if (std::sscanf(line, " %63[A-Za-z_] = %llu", key, &v) != 2)
            continue;
        if (!std::strcmp(key, "k")) {
            *k = (unsigned)v;
            have_k = true;
        } else if (!std::strcmp(key, "hash_num")) {
            *hash_num = (unsigned)v;
            have_h = true;
        } else if (!std::strcmp(key, "bytes")) {
            *bytes = v;
            have_b = true;
        }
    }
    if (!ended || !have_k || !have_h || !have_b)
        return false;
    const long at = std::ftell(f);
    std::fseek(f, 0, SEEK_END);
    const long end = std::ftell(f);
    std::fseek(f, at, SEEK_SET);
    return (unsigned long long)(end - at) == *bytes;
}

int gp_filter_load(const char* path, uint32_t* k, uint32_t* hash_num, uint64_t* bytes, uint8_t* buf,
                   uint64_t buf_bytes)
try {
    if (!path)
        return GP_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "rb");
    if (!f) {
        set_error(std::string("cannot open ") + path);
        return GP_ERR_INVALID_ARG;
    }
    unsigned kk = 0, hh = 0;
    unsigned long long bb = 0;
    char magic[8] = { 0 };
    if (std::fscanf(f, "%7s %u %u %llu", magic, &kk, &hh, &bb) != 4 || std::strcmp(magic, "GPBF1") != 0 ||
        std::fgetc(f) != '\n') {
        std::rewind(f);
        if (!read_btl_header(f, &kk, &hh, &bb)) {
            std::fclose(f);
            set_error(std::string(path) + " is neither a GPBF1 file nor a [BTLKmerBloomFilter_v*] file with bytes, hash_num and k");
            return GP_ERR_UNSUPPORTED;
        }
    }
    if (k)
        *k = kk;
    if (hash_num)
        *hash_num = hh;
    if (bytes)
        *bytes = bb;
    int rc = GP_OK;
    if (buf) {
        if (buf_bytes < bb) {
            set_error("gp_filter_load: buffer too small");
            rc = GP_ERR_INVALID_ARG;
        } else if (std::fread(buf, 1, bb, f) != bb) {
            set_error(std::string("short read from ") + path);
            rc = GP_ERR_INVALID_ARG;
        }
    }
    std::fclose(f);
    return rc;
} GP_CATCH("gp_filter_load")

/* What serve_batch does before it fills the filters (src/goldpolish_targeted_bfs.cpp:86-131), for
 * every target of every batch, plus what scripts/goldpolish:344-355 does when it writes
 * batch.fa: the contig comes from the target file, its reads are the mapped reads ranked by
 * truncated average phred (ties by id), cut to len * s / 10000, their bases give the counting
 * threshold. */
int gp_job_create_from_files(gp_ctx* ctx, const gp_params* params, const gp_seqindex* targets,
                             const gp_seqindex* reads, gp_mappings* mappings,
                             double max_reads_per_10kbp, uint32_t n_batches,
                             const uint32_t* batch_target_begin, const char* const* target_ids,
                             gp_job** out)
try {
    if (!ctx || !params || !targets || !reads || !mappings || !batch_target_begin || !target_ids || !out)
        return GP_ERR_INVALID_ARG;
    const uint32_t n_contigs = batch_target_begin[n_batches];
    std::vector<uint64_t> contig_off(n_contigs + 1, 0), read_off(1, 0);
    std::vector<uint32_t> name_len(n_contigs), read_begin(n_contigs + 1, 0), thr(n_contigs);
    std::vector<uint8_t> contig_buf, read_buf;
    std::vector<std::string> headers(n_contigs);
    for (uint32_t c = 0; c < n_contigs; c++) {
        const auto* t = targets->find(target_ids[c]);
        if (!t) {
            set_error(std::string("target not in index: ") + target_ids[c]);
            return GP_ERR_NOT_FOUND;
        }
        contig_buf.resize(contig_off[c] + t->len);
        int rc = read_exact(targets, t->start, t->len, contig_buf.data() + contig_off[c]);
        if (rc != GP_OK)
            return rc;
        contig_off[c + 1] = contig_off[c] + t->len;
        {
            // header line = the bytes between the previous newline and the sequence: ">id[ comment]"
            const uint64_t nl = t->start ? t->start - 1 : 0; // the newline that ends the header
            uint64_t hb = nl;
            // walk back to the previous newline (or the start of the file), one pread per 4 KiB
            for (bool found = false; hb > 0 && !found;) {
                uint8_t chunk[4096];
                const uint64_t n = std::min<uint64_t>(hb, sizeof chunk);
                if (pread(targets->fd, chunk, n, (off_t)(hb - n)) != (ssize_t)n) {
                    set_error(std::string("cannot read the header of target ") + target_ids[c]);
                    return GP_ERR_INVALID_ARG;
                }
                uint64_t i = n;
                while (i > 0 && chunk[i - 1] != '\n')
                    i--;
                found = i > 0;
                hb -= n - i;
            }
            name_len[c] = nl > hb ? (uint32_t)(nl - hb - 1) : 0; // without the '>'
            headers[c].resize(name_len[c]);
            if (name_len[c] && read_exact(targets, hb + 1, name_len[c], (uint8_t*)&headers[c][0]) != GP_OK)
                return GP_ERR_INVALID_ARG;
        }
        uint64_t n_ids = 0;
        const char* const* ids = nullptr;
        gp_mappings_get(mappings, target_ids[c], &n_ids, &ids);
        std::vector<const char*> cid(ids, ids + n_ids);
        std::vector<double> phred(n_ids);
        std::vector<uint64_t> lens(n_ids);
        for (uint64_t i = 0; i < n_ids; i++) {
            const auto* r = reads->find(cid[i]);
            if (!r) {
                set_error(std::string("mapped read not in the read index: ") + cid[i]);
                return GP_ERR_NOT_FOUND;
            }
            phred[i] = r->phred;
            lens[i] = r->len;
        }
        std::vector<uint32_t> order(n_ids ? n_ids : 1);
        uint64_t bases = 0;
        const size_t n_sel = gp_rank_reads(n_ids, cid.data(), phred.data(), lens.data(), t->len,
                                           max_reads_per_10kbp, order.data(), &bases);
        thr[c] = (uint32_t)gp_kmer_threshold(bases);
        for (size_t i = 0; i < n_sel; i++) {
            const auto* r = reads->find(cid[order[i]]);
            const uint64_t at = read_off.back();
            read_buf.resize(at + r->len);
            if ((rc = read_exact(reads, r->start, r->len, read_buf.data() + at)) != GP_OK)
                return rc;
            read_off.push_back(at + r->len);
        }
        read_begin[c + 1] = (uint32_t)(read_off.size() - 1);
    }
    gp_batch_set bs{};
    bs.n_batches = n_batches;
    bs.batch_contig_begin = batch_target_begin;
    bs.n_contigs = n_contigs;
    bs.contig_buf = contig_buf.data();
    bs.contig_off = contig_off.data();
    bs.contig_name_len = name_len.data();
    bs.contig_read_begin = read_begin.data();
    bs.contig_kmer_threshold = thr.data();
    bs.n_reads = (uint32_t)(read_off.size() - 1);
    bs.read_buf = read_buf.data();
    bs.read_off = read_off.data();
    const int rc_create = gp_job_create(ctx, params, &bs, out);
    if (rc_create == GP_OK)
        (*out)->headers = std::move(headers);
    return rc_create;
} GP_CATCH("gp_job_create_from_files")

/* The batch's final FASTA (scripts/goldpolish-make:46-47 `goldpolish-to-upper`, collected in input
 * order by scripts/goldpolish-reaper:51-73): ">id[ comment]" and the polished sequence, records
 * ntEdit dropped left out.  names == NULL: the header lines of the target file (jobs made by
 * gp_job_create_from_files). */
int gp_job_write_fasta(gp_job* job, const char* const* names, const char* path, int append)
try {
    if (!job || !path)
        return GP_ERR_INVALID_ARG;
    if (!names && job->headers.size() != job->n_contigs) {
        set_error("gp_job_write_fasta: names are needed for a job that was not made from files");
        return GP_ERR_INVALID_ARG;
    }
    std::vector<uint64_t> lens(job->n_contigs ? job->n_contigs : 1);
    std::vector<uint8_t> dropped(lens.size());
    uint64_t total = 0;
    int rc = gp_job_get_lengths(job, lens.data(), dropped.data(), &total);
    if (rc != GP_OK)
        return rc;
    std::vector<uint8_t> buf(total ? total : 1);
    if ((rc = gp_job_get_sequences(job, buf.data(), buf.size())) != GP_OK)
        return rc;
    FILE* f = std::fopen(path, append ? "ab" : "wb");
    if (!f) {
        set_error(std::string("cannot open ") + path + " for writing: " + std::strerror(errno));
        return GP_ERR_IO;
    }
    // GoldPolish polishes in /dev/shm, where "no space left" is the usual failure: every write is
    // checked, and a file that could not be written whole is put back to what it was before
    // (a new file is removed, an appended one cut back), so that no truncated batch is collected
    // (the size of the file, not ftell(): the position of a freshly opened append stream is
    // implementation-defined -- glibc reports the size, musl 0, and cutting back to 0 would
    // destroy the batches already collected)
    long long before = 0;
    if (append) {
        struct stat sb;
        before = fstat(fileno(f), &sb) == 0 ? (long long)sb.st_size : -1;
    }
    bool ok = true;
    uint64_t at = 0;
    for (uint32_t c = 0; ok && c < job->n_contigs; c++) {
        if (dropped[c])
            continue;
        const char* nm = names ? names[c] : job->headers[c].c_str();
        ok = std::fputc('>', f) != EOF && std::fputs(nm, f) != EOF && std::fputc('\n', f) != EOF &&
             std::fwrite(buf.data() + at, 1, lens[c], f) == lens[c] && std::fputc('\n', f) != EOF;
        at += lens[c];
    }
    ok = ok && std::fflush(f) == 0;
    const int write_errno = errno;
    bool cut_back = true;
    if (!ok && append)
        cut_back = before >= 0 && ftruncate(fileno(f), (off_t)before) == 0;
    if (std::fclose(f) != 0 && ok) {
        set_error(std::string("closing ") + path + " failed: " + std::strerror(errno));
        if (!append)
            (void)std::remove(path);
        return GP_ERR_IO;
    }
    if (!ok) {
        if (!append)
            (void)std::remove(path);
        set_error(std::string("writing ") + path + " failed: " + std::strerror(write_errno) +
                  (cut_back ? "" : "; the file could NOT be cut back to its previous size and holds a partial batch"));
        return GP_ERR_IO;
    }
    return GP_OK;
} GP_CATCH("gp_job_write_fasta")

} // extern "C"
