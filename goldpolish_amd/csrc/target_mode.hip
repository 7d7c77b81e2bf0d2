// target_mode.hip -- GoldPolish-Target's three file-to-file steps around the polishing path,
// on the host.  Plain C++ (no kernels); needed to run a "--target" job
// without snakemake / intervaltree / btllib's Python module.
//
// What it replaces in the reference:
//   * gp_target_extract     /root/reference/scripts/goldpolish-target-extract-seq.py:60-148
//   * gp_target_update_paf  /root/reference/scripts/goldpolish-target-update-mapping.py:70-155
//   * gp_target_reinsert    /root/reference/scripts/goldpolish-target-post-processing.py:28-94
// as chained by scripts/goldpolish-target-run-pipeline.smk:81-111.  The first and the last are
// pinned by the reference's own test data (tests/golden/target/, copied from
// /root/reference/tests/ and tests/expected_files/).
//
// Kept on purpose: window sequences are written upper-cased with the comment "start-end"; an
// upper-case island shorter than two flanks is absorbed into the soft-masked run before it;
// soft-masked runs of a single base are not windows; re-insertion trusts record order and
// drops the last base of a contig when a window ends exactly one base before its end
// (`start < len - 1`); the rewritten PAF rows end in CR LF (Python's csv.writer) and keep only
// the twelve standard columns.
#include "gp_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace gp {
namespace {

struct Record {
    std::string id, comment, seq;
};

// btllib::SeqReader, LONG_MODE, FASTA: header = id up to the first blank, the rest is the
// comment; sequence lines are concatenated
int read_fasta(const std::string& path, std::vector<Record>& out)
{
    std::ifstream in(path);
    if (!in) {
        set_error("cannot open " + path);
        return GP_ERR_INVALID_ARG;
    }
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r')
            line.pop_back();
        if (!line.empty() && line[0] == '>') {
            Record r;
            size_t p = line.find_first_of(" \t", 1);
            r.id = line.substr(1, p == std::string::npos ? std::string::npos : p - 1);
            if (p != std::string::npos) {
                size_t q = line.find_first_not_of(" \t", p);
                if (q != std::string::npos)
                    r.comment = line.substr(q);
            }
            out.push_back(std::move(r));
        } else if (!out.empty()) {
            out.back().seq += line;
        }
    }
    return GP_OK;
}

// btllib::SeqWriter::write(id, comment, seq), FASTA
void write_record(FILE* f, const std::string& id, const std::string& comment, const std::string& seq)
{
    std::fputc('>', f);
    std::fwrite(id.data(), 1, id.size(), f);
    if (!comment.empty()) {
        std::fputc(' ', f);
        std::fwrite(comment.data(), 1, comment.size(), f);
    }
    std::fputc('\n', f);
    std::fwrite(seq.data(), 1, seq.size(), f);
    std::fputc('\n', f);
}

std::string upper(std::string s)
{
    for (auto& c : s)
        if (c >= 'a' && c <= 'z')
            c = (char)(c - 32);
    return s;
}

// python: sequence[a:b] for 0 <= a, any b
std::string slice(const std::string& s, long a, long b)
{
    const long n = (long)s.size();
    a = std::min(std::max(a, 0L), n);
    b = std::min(std::max(b, 0L), n);
    return b > a ? s.substr((size_t)a, (size_t)(b - a)) : std::string();
}

struct Run {
    bool lower;
    size_t len;
};

// extract_masked_subsequences, goldpolish-target-extract-seq.py:60-108
void extract_masked(const Record& r, long flank, FILE* out)
{
    // re.findall(r"([A-Z]+|[a-z]+)"): maximal runs of one case; anything else only separates
    std::vector<Run> runs;
    for (size_t i = 0; i < r.seq.size();) {
        const char c = r.seq[i];
        const bool up = c >= 'A' && c <= 'Z', lo = c >= 'a' && c <= 'z';
        if (!up && !lo) {
            i++;
            continue;
        }
        size_t j = i;
        while (j < r.seq.size() && ((up && r.seq[j] >= 'A' && r.seq[j] <= 'Z') ||
                                    (lo && r.seq[j] >= 'a' && r.seq[j] <= 'z')))
            j++;
        runs.push_back({ lo, j - i });
        i = j;
    }
    if (runs.empty())
        return; // the reference raises IndexError on a record without letters
    std::vector<Run> kept;
    kept.push_back(runs[0]);
    for (size_t i = 1; i < runs.size(); i++) {
        const Run& s = runs[i];
        if (!s.lower) {
            if ((long)s.len >= 2 * flank)
                kept.push_back(s);
            else
                kept.back().len += s.len; // previous run is soft-masked (runs alternate): absorbed, lower-cased
        } else if (!kept.back().lower) {
            kept.push_back(s);
        } else {
            kept.back().len += s.len;
        }
    }
    long idx = 0, gap_count = 1;
    for (const Run& s : kept) {
        if (s.lower && s.len > 1) { // MIN_GAP_LENGTH = 1
            const long start = std::max(0L, idx - flank);
            const long end = std::min((long)r.seq.size(), idx + (long)s.len + flank);
            if (end > start) {
                write_record(out, r.id + "." + std::to_string(gap_count),
                             std::to_string(start) + "-" + std::to_string(end), upper(slice(r.seq, start, end)));
                gap_count++;
            }
        }
        idx += (long)s.len;
    }
}

struct Coord {
    std::string start, end;
};

// extract_subsequences_from_bed, goldpolish-target-extract-seq.py:111-143
void extract_bed(const Record& r, long flank, FILE* out, const std::vector<Coord>& coords)
{
    std::vector<Coord> kept;
    kept.push_back(coords[0]);
    for (size_t i = 1; i < coords.size(); i++) {
        if (std::stol(coords[i].start) - std::stol(kept.back().end) < 2 * flank)
            kept.back().end = coords[i].end;
        else
            kept.push_back(coords[i]);
    }
    long count = 0;
    for (const Coord& c : kept) {
        const long start = std::max(0L, std::stol(c.start) - flank);
        const long end = std::min((long)r.seq.size(), std::stol(c.end) + flank);
        count++;
        write_record(out, r.id + "." + std::to_string(count), std::to_string(start) + "-" + std::to_string(end),
                     upper(slice(r.seq, start, end)));
    }
}

std::vector<std::string> split_tabs(const std::string& line)
{
    std::vector<std::string> out;
    size_t a = 0;
    for (;;) {
        size_t b = line.find('\t', a);
        out.push_back(line.substr(a, b == std::string::npos ? std::string::npos : b - a));
        if (b == std::string::npos)
            break;
        a = b + 1;
    }
    return out;
}

std::string before_last_dot(const std::string& s)
{
    const size_t p = s.rfind('.');
    return p == std::string::npos ? s : s.substr(0, p);
}

std::string after_last_dot(const std::string& s)
{
    const size_t p = s.rfind('.');
    return p == std::string::npos ? s : s.substr(p + 1);
}

} // namespace
} // namespace gp

using namespace gp;

extern "C" {

int gp_target_extract(const char* fasta_path, const char* bed_path, uint32_t flank, const char* out_path)
try {
    if (!fasta_path || !out_path)
        return GP_ERR_INVALID_ARG;
    std::vector<Record> recs;
    int rc = read_fasta(fasta_path, recs);
    if (rc != GP_OK)
        return rc;
    std::unordered_map<std::string, std::vector<Coord>> bed;
    if (bed_path) {
        std::ifstream in(bed_path);
        if (!in) {
            set_error(std::string("cannot open ") + bed_path);
            return GP_ERR_INVALID_ARG;
        }
        std::string line;
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r')
                line.pop_back();
            const auto cols = split_tabs(line);
            if (cols.size() < 3) {
                set_error(std::string("BED row with fewer than three columns in ") + bed_path);
                return GP_ERR_INVALID_ARG;
            }
            bed[cols[0]].push_back({ cols[1], cols[2] });
        }
    }
    FILE* out = std::fopen(out_path, "w");
    if (!out) {
        set_error(std::string("cannot write ") + out_path);
        return GP_ERR_INVALID_ARG;
    }
    try {
        for (const Record& r : recs) {
            if (!bed_path) {
                extract_masked(r, (long)flank, out);
            } else {
                auto it = bed.find(r.id);
                if (it != bed.end())
                    extract_bed(r, (long)flank, out, it->second);
            }
        }
    } catch (const std::exception&) {
        std::fclose(out);
        set_error(std::string("malformed coordinate in ") + bed_path);
        return GP_ERR_INVALID_ARG;
    }
    std::fclose(out);
    return GP_OK;
} GP_CATCH("gp_target_extract")

int gp_target_update_paf(const char* gaps_fasta_path, const char* paf_path, const char* out_path)
try {
    if (!gaps_fasta_path || !paf_path || !out_path)
        return GP_ERR_INVALID_ARG;
    struct Window {
        long begin, end; // [begin, end): the comment's end is inclusive here (+1, :74)
        std::string name;
    };
    std::vector<Record> gaps;
    int rc = read_fasta(gaps_fasta_path, gaps);
    if (rc != GP_OK)
        return rc;
    std::unordered_map<std::string, std::vector<Window>> windows;
    try {
        for (const Record& g : gaps) {
            const size_t dash = g.comment.find('-');
            if (dash == std::string::npos)
                throw std::invalid_argument("comment");
            // split("-")[1]: up to the next dash, if any
            const std::string second = g.comment.substr(dash + 1, g.comment.find('-', dash + 1) - dash - 1);
            windows[before_last_dot(g.id)].push_back({ std::stol(g.comment.substr(0, dash)),
                                                       std::stol(second) + 1, g.id });
        }
    } catch (const std::exception&) {
        set_error(std::string("window without a start-end comment in ") + gaps_fasta_path);
        return GP_ERR_INVALID_ARG;
    }
    std::ifstream in(paf_path);
    if (!in) {
        set_error(std::string("cannot open ") + paf_path);
        return GP_ERR_INVALID_ARG;
    }
    FILE* out = std::fopen(out_path, "w");
    if (!out) {
        set_error(std::string("cannot write ") + out_path);
        return GP_ERR_INVALID_ARG;
    }
    std::string line;
    try {
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r')
                line.pop_back();
            if (line.empty())
                continue;
            const auto c = split_tabs(line);
            if (c.size() < 12)
                throw std::invalid_argument("columns");
            long q_len = std::stol(c[1]), q_start = std::stol(c[2]), q_end = std::stol(c[3]);
            long t_len = std::stol(c[6]), t_start = std::stol(c[7]), t_end = std::stol(c[8]);
            const long matches = std::stol(c[9]), quality = std::stol(c[11]);
            (void)std::stol(c[10]);
            auto it = windows.find(c[5]);
            if (it == windows.end())
                continue;
            // tree.overlap(start, end); with several hits the reference takes whichever its set
            // yields first (and warns) -- here: the leftmost window
            const Window* hit = nullptr;
            for (const Window& w : it->second)
                if (w.begin < t_end && w.end > t_start && (!hit || w.begin < hit->begin))
                    hit = &w;
            if (!hit)
                continue;
            long nt_start, nt_end, nq_start = q_start, nq_end = q_end;
            if (hit->begin < t_start && hit->end > t_end) {
                nt_start = t_start - hit->begin;
                nt_end = t_end - hit->begin;
            } else if (hit->begin < t_start && hit->end <= t_end) {
                nt_start = t_start - hit->begin;
                nt_end = hit->end - hit->begin;
                nq_end = q_end - (t_end - hit->end);
            } else if (hit->begin >= t_start && hit->end > t_end) {
                nt_start = 0;
                nt_end = t_end - hit->begin;
                nq_start = q_start + (hit->begin - t_start);
            } else {
                nt_start = 0;
                nt_end = hit->end - hit->begin;
                nq_start = q_start + (hit->begin - t_start);
                nq_end = q_end - (t_end - hit->end);
            }
            std::fprintf(out, "%s\t%ld\t%ld\t%ld\t%s\t%s\t%ld\t%ld\t%ld\t%ld\t%ld\t%ld\r\n", c[0].c_str(), q_len,
                         nq_start, nq_end, c[4].c_str(), hit->name.c_str(), t_len, nt_start, nt_end, matches,
                         nt_end - nt_start, quality);
        }
    } catch (const std::exception&) {
        std::fclose(out);
        set_error(std::string("malformed PAF row in ") + paf_path);
        return GP_ERR_INVALID_ARG;
    }
    std::fclose(out);
    return GP_OK;
} GP_CATCH("gp_target_update_paf")

int gp_target_reinsert(const char* fasta_path, const char* polished_gaps_path, const char* out_path)
try {
    if (!fasta_path || !polished_gaps_path || !out_path)
        return GP_ERR_INVALID_ARG;
    std::vector<Record> recs, gaps;
    int rc = read_fasta(fasta_path, recs);
    if (rc == GP_OK)
        rc = read_fasta(polished_gaps_path, gaps);
    if (rc != GP_OK)
        return rc;
    // make_seq_dict: contig -> number of its last window (as text)
    std::unordered_map<std::string, std::string> last_window;
    for (const Record& g : gaps)
        last_window[before_last_dot(g.id)] = after_last_dot(g.id);
    FILE* out = std::fopen(out_path, "w");
    if (!out) {
        set_error(std::string("cannot write ") + out_path);
        return GP_ERR_INVALID_ARG;
    }
    size_t gi = 0; // the gap reader's position
    try {
        for (const Record& r : recs) {
            auto it = last_window.find(r.id);
            if (it == last_window.end()) {
                write_record(out, r.id, "", r.seq);
                continue;
            }
            if (gi >= gaps.size())
                throw std::out_of_range("gaps");
            const long last = std::stol(it->second);
            std::string gap_name = before_last_dot(gaps[gi].id) + ".0";
            std::string updated;
            long start = 0;
            while (before_last_dot(gap_name) == r.id && std::stol(after_last_dot(gap_name)) < last) {
                if (gi >= gaps.size())
                    throw std::out_of_range("gaps");
                const Record& g = gaps[gi++];
                gap_name = g.id;
                const size_t dash = g.comment.find('-');
                if (dash == std::string::npos)
                    throw std::invalid_argument("comment");
                const std::string second = g.comment.substr(dash + 1, g.comment.find('-', dash + 1) - dash - 1);
                const long g_start = std::stol(g.comment.substr(0, dash)), g_end = std::stol(second);
                updated += slice(r.seq, start, g_start); // insert_seq, :28-46
                updated += g.seq;
                start = g_end;
            }
            if (start < (long)r.seq.size() - 1)
                updated += slice(r.seq, start, (long)r.seq.size());
            write_record(out, r.id, "", updated);
        }
    } catch (const std::exception&) {
        std::fclose(out);
        set_error(std::string("windows in ") + polished_gaps_path + " do not follow the order of " + fasta_path);
        return GP_ERR_INVALID_ARG;
    }
    std::fclose(out);
    return GP_OK;
} GP_CATCH("gp_target_reinsert")

} // extern "C"
