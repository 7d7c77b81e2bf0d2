/*
 * oracle/sealer.c -- CPU ORACLE (test infrastructure, never shipped).
 * Stage C of the path: Sealer gap filling = Konnector's constrained
 * bidirectional BFS in the Bloom-filter de Bruijn graph, restated from
 *   /root/reference/subprojects/sealer/sealer.cpp
 *       findFlanks :695-733, makePseudoReads :560-577, kRun :579-685,
 *       merge :425-517, insertIntoScaffold :525-558, parameters :939-957
 *   Konnector/konnector.h connectPairs :234-438
 *   Konnector/DBGBloomAlgorithms.h getStartKmerPos :51-108, pathToSeq :21-30,
 *       maskNew :183-210
 *   Konnector/DBGBloom.h neighbour order A,C,G,T :107-118,176-183,243-250;
 *       vertex_exists :295-300
 *   Graph/BidirectionalBFS.h :14-156
 *   Graph/ConstrainedBidiBFSVisitor.h :98-142,160-221,306-492
 *   Graph/AllPathsSearch.h :28-104, Graph/HashGraph.h add_edge :125-144
 *   Align/alignGlobal.cc :38-198, Align/alignGlobal.h :47-83
 *   Common/Sequence.cpp complement :21-47, ambiguity tables :141-204
 * PARITY UNPINNED (needs boost + btllib to build; the reference's tests for it
 * are end-to-end only).  Two points are *defined* here because the reference
 * leaves them to library internals:
 *   - common edges are visited in insertion order (the reference iterates a
 *     boost::unordered_set, whose order depends on the boost version); this
 *     only matters when more than one connecting path exists;
 *   - a flank k-mer holding a byte outside ACGT counts as "not in the graph"
 *     (btllib NtHash skips such k-mers).
 * The 500 MB traversal memory check (ConstrainedBidiBFSVisitor.h:342-354) is
 * not modelled: with max-cost 100000 it cannot be reached.
 */
#include "gpo.h"

#include <ctype.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define NO_LIMIT UINT_MAX
#define NO_MATCH UINT_MAX

enum { FWD = 0, REV = 1 };
enum { R_FOUND = 0, R_TOO_MANY_PATHS, R_TOO_MANY_BRANCHES, R_CYCLE, R_MAX_COST, R_MEM, R_NO_PATH };
enum { V_SUCCESS = 0, V_SKIP, V_ABORT };

/* ------------------------------------------------------------------ utils */
static char comp_base(char c)
{
    char r;
    switch (toupper((unsigned char)c)) {
    case 'A': r = 'T'; break;
    case 'C': r = 'G'; break;
    case 'G': r = 'C'; break;
    case 'T': r = 'A'; break;
    case 'M': r = 'K'; break;
    case 'R': r = 'Y'; break;
    case 'W': r = 'W'; break;
    case 'S': r = 'S'; break;
    case 'Y': r = 'R'; break;
    case 'K': r = 'M'; break;
    case 'V': r = 'B'; break;
    case 'H': r = 'D'; break;
    case 'D': r = 'H'; break;
    case 'B': r = 'V'; break;
    default: r = 'N'; break; /* the reference aborts on other bytes */
    }
    return islower((unsigned char)c) ? (char)tolower((unsigned char)r) : r;
}

static void revcomp(const char* s, size_t n, char* out)
{
    for (size_t i = 0; i < n; i++)
        out[i] = comp_base(s[n - 1 - i]);
    out[n] = 0;
}

typedef struct {
    const uint8_t* bf;
    size_t bf_bytes;
    unsigned h, k;
    uint64_t tests;
} graph_t;

static int kmer_in_graph(graph_t* g, const char* kmer)
{
    g->tests++;
    for (unsigned i = 0; i < g->k; i++)
        if (gpo_seed((unsigned char)kmer[i]) == 0)
            return 0;
    uint64_t hv[16];
    gpo_extend(gpo_ntf64(kmer, g->k) + gpo_ntr64(kmer, g->k), g->k, g->h, hv);
    return gpo_bf_contains(g->bf, g->bf_bytes, hv, g->h);
}

/* DBGBloomAlgorithms.h:51-108, dir == FORWARD only (both call sites) */
static unsigned start_kmer_pos(graph_t* g, const char* seq, size_t len, unsigned need)
{
    const unsigned k = g->k;
    if (len < k)
        return NO_MATCH;
    unsigned run = 0, best_len = 0, best_pos = 0;
    for (long pos = (long)(len - k); pos >= 0; pos--) {
        if (!kmer_in_graph(g, seq + pos)) {
            if (run > best_len) {
                best_pos = (unsigned)(pos + 1);
                best_len = run;
            }
            run = 0;
        } else {
            run++;
            if (run >= need)
                return (unsigned)pos;
        }
    }
    if (run > best_len) {
        best_pos = 0;
        best_len = run;
    }
    return best_len == 0 ? NO_MATCH : best_pos;
}

/* ------------------------------------------------------- vertex bookkeeping */
typedef struct {
    uint32_t* v;
    uint32_t n, cap;
} u32vec;

static void u32_push(u32vec* a, uint32_t x)
{
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 4;
        a->v = (uint32_t*)realloc(a->v, a->cap * sizeof(uint32_t));
    }
    a->v[a->n++] = x;
}

typedef struct {
    uint8_t color[2];    /* 0 white, 1 gray, 2 black */
    uint16_t depth[2];   /* DepthMap: operator[] default-inserts 0 */
    uint8_t in_graph[2]; /* vertex present in traversal history graph */
    u32vec succ[2];      /* successor lists, insertion order, no duplicates */
} vert_t;

typedef struct {
    unsigned k;
    char* arena; /* k bytes per vertex */
    vert_t* vt;
    uint32_t n, cap;
    uint32_t* slots; /* open addressing, value = id+1 */
    uint32_t nslots;
} vtab_t;

static uint64_t str_hash(const char* s, unsigned k)
{
    uint64_t h = 1469598103934665603ULL;
    for (unsigned i = 0; i < k; i++) {
        h ^= (unsigned char)s[i];
        h *= 1099511628211ULL;
    }
    return h;
}

static void vtab_init(vtab_t* t, unsigned k)
{
    memset(t, 0, sizeof *t);
    t->k = k;
    t->nslots = 1024;
    t->slots = (uint32_t*)calloc(t->nslots, sizeof(uint32_t));
}

static void vtab_free(vtab_t* t)
{
    for (uint32_t i = 0; i < t->n; i++) {
        free(t->vt[i].succ[0].v);
        free(t->vt[i].succ[1].v);
    }
    free(t->arena);
    free(t->vt);
    free(t->slots);
}

static void vtab_rehash(vtab_t* t)
{
    free(t->slots);
    t->nslots *= 4;
    t->slots = (uint32_t*)calloc(t->nslots, sizeof(uint32_t));
    for (uint32_t i = 0; i < t->n; i++) {
        uint64_t s = str_hash(t->arena + (size_t)i * t->k, t->k) & (t->nslots - 1);
        while (t->slots[s])
            s = (s + 1) & (t->nslots - 1);
        t->slots[s] = i + 1;
    }
}

static uint32_t vtab_get(vtab_t* t, const char* kmer)
{
    uint64_t s = str_hash(kmer, t->k) & (t->nslots - 1);
    while (t->slots[s]) {
        uint32_t id = t->slots[s] - 1;
        if (memcmp(t->arena + (size_t)id * t->k, kmer, t->k) == 0)
            return id;
        s = (s + 1) & (t->nslots - 1);
    }
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 256;
        t->arena = (char*)realloc(t->arena, (size_t)t->cap * t->k);
        t->vt = (vert_t*)realloc(t->vt, (size_t)t->cap * sizeof(vert_t));
    }
    uint32_t id = t->n++;
    memcpy(t->arena + (size_t)id * t->k, kmer, t->k);
    memset(&t->vt[id], 0, sizeof(vert_t));
    t->slots[s] = id + 1;
    if ((uint64_t)t->n * 2 > t->nslots)
        vtab_rehash(t);
    return id;
}

static const char* vtab_kmer(const vtab_t* t, uint32_t id) { return t->arena + (size_t)id * t->k; }

/* HashGraph::add_edge (Graph/HashGraph.h:125-144) */
static void hist_add_edge(vtab_t* t, int gi, uint32_t u, uint32_t v)
{
    vert_t* U = &t->vt[u];
    U->in_graph[gi] = 1;
    int dup = 0;
    for (uint32_t i = 0; i < U->succ[gi].n; i++)
        if (U->succ[gi].v[i] == v) {
            dup = 1;
            break;
        }
    if (!dup)
        u32_push(&U->succ[gi], v);
    t->vt[v].in_graph[gi] = 1;
}

/* ---------------------------------------------------------------- search */
typedef struct {
    uint32_t* v;
    uint32_t n;
} path_t;

typedef struct {
    path_t* p;
    uint32_t n, cap;
} pathlist_t;

static void pl_push(pathlist_t* l, const uint32_t* v, uint32_t n)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 4;
        l->p = (path_t*)realloc(l->p, l->cap * sizeof(path_t));
    }
    l->p[l->n].v = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    memcpy(l->p[l->n].v, v, n * sizeof(uint32_t));
    l->p[l->n].n = n;
    l->n++;
}

static void pl_free(pathlist_t* l)
{
    for (uint32_t i = 0; i < l->n; i++)
        free(l->p[i].v);
    free(l->p);
    memset(l, 0, sizeof *l);
}

typedef struct {
    graph_t* g;
    vtab_t vt;
    uint32_t start, goal;
    unsigned max_paths, max_branches, max_cost, cost;
    uint16_t min_len, max_len, max_depth[2];
    int too_many_branches, max_cost_exceeded, too_many_paths;
    uint32_t ce_u[64], ce_v[64], n_ce; /* common edges, insertion order */
    pathlist_t found;
} search_t;

/* Graph/AllPathsSearch.h:28-104 over history graph gi */
static int all_paths(search_t* s, int gi, uint32_t from, uint32_t to, unsigned max_paths,
                     unsigned max_depth, unsigned max_cost, pathlist_t* out, unsigned* cost_out)
{
    vtab_t* t = &s->vt;
    uint32_t cap = 64, n = 0;
    uint32_t* path = (uint32_t*)malloc(cap * sizeof(uint32_t));
    uint32_t* it = (uint32_t*)malloc(cap * sizeof(uint32_t));
    uint8_t* visited = (uint8_t*)calloc(t->n, 1);
    uint8_t* cyc = (uint8_t*)calloc(t->n, 1);
    uint32_t n_cyc = 0;
    unsigned cost = 0;
    int rc = -1;

    path[0] = from;
    it[0] = 0;
    n = 1;
    visited[from] = 1;

    while (n > 0 && cost <= max_cost) {
        if (path[n - 1] == to) {
            if (max_paths != NO_LIMIT && out->n >= max_paths) {
                rc = R_TOO_MANY_PATHS;
                break;
            }
            if (n_cyc) {
                rc = R_CYCLE;
                break;
            }
            pl_push(out, path, n);
        }
        while (n > 0) {
            uint32_t cur = path[n - 1];
            const u32vec* sv = &t->vt[cur].succ[gi];
            if ((max_depth != NO_LIMIT && (n - 1) >= max_depth) || it[n - 1] == sv->n) {
                visited[cur] = 0;
                if (cyc[cur]) {
                    cyc[cur] = 0;
                    n_cyc--;
                }
                n--;
                if (n > 0)
                    it[n - 1]++;
            } else {
                uint32_t v = sv->v[it[n - 1]];
                if (visited[v]) {
                    if (!cyc[v]) {
                        cyc[v] = 1;
                        n_cyc++;
                    }
                    it[n - 1]++;
                } else {
                    if (n == cap) {
                        cap *= 2;
                        path = (uint32_t*)realloc(path, cap * sizeof(uint32_t));
                        it = (uint32_t*)realloc(it, cap * sizeof(uint32_t));
                    }
                    path[n] = v;
                    it[n] = 0;
                    n++;
                    visited[v] = 1;
                    cost++;
                    break;
                }
            }
        }
    }
    if (rc < 0) {
        if (cost > max_cost)
            rc = R_MAX_COST;
        else if (out->n == 0)
            rc = R_NO_PATH;
        else
            rc = R_FOUND;
    }
    *cost_out = cost;
    free(path);
    free(it);
    free(visited);
    free(cyc);
    return rc;
}

/* ConstrainedBidiBFSVisitor.h:427-492 */
static int build_paths_edge(search_t* s, uint32_t u, uint32_t v)
{
    if (s->cost > s->max_cost) {
        s->max_cost_exceeded = 1;
        return R_MAX_COST;
    }
    pathlist_t left = { 0, 0, 0 }, right = { 0, 0, 0 };
    unsigned c = 0;
    unsigned max_to_start = s->max_paths - s->found.n;
    int rc = all_paths(s, FWD, u, s->start, max_to_start, s->max_depth[FWD],
                       s->max_cost - s->cost, &left, &c);
    s->cost += c;
    if (rc == R_FOUND) {
        unsigned max_to_goal = (s->max_paths - s->found.n) / left.n;
        rc = all_paths(s, REV, v, s->goal, max_to_goal, s->max_depth[REV],
                       s->max_cost - s->cost, &right, &c);
        s->cost += c;
        if (rc == R_FOUND) {
            int added = 0;
            for (uint32_t a = 0; a < left.n; a++)
                for (uint32_t b = 0; b < right.n; b++) {
                    uint32_t tot = left.p[a].n + right.p[b].n;
                    if (tot < s->min_len || tot > s->max_len)
                        continue;
                    uint32_t* m = (uint32_t*)malloc(tot * sizeof(uint32_t));
                    for (uint32_t i = 0; i < left.p[a].n; i++)
                        m[i] = left.p[a].v[left.p[a].n - 1 - i];
                    memcpy(m + left.p[a].n, right.p[b].v, right.p[b].n * sizeof(uint32_t));
                    pl_push(&s->found, m, tot);
                    free(m);
                    added = 1;
                }
            rc = added ? R_FOUND : R_NO_PATH;
        }
    }
    if (rc == R_MAX_COST)
        s->max_cost_exceeded = 1;
    else if (rc == R_TOO_MANY_PATHS)
        s->too_many_paths = 1;
    pl_free(&left);
    pl_free(&right);
    return rc;
}

static int bump_cost(search_t* s)
{
    if (s->cost >= s->max_cost) {
        s->max_cost_exceeded = 1;
        return V_ABORT;
    }
    s->cost++;
    return V_SUCCESS;
}

/* recordEdgeTraversal: history graphs store the edge pointing back to the
 * traversal's origin (ConstrainedBidiBFSVisitor.h:360-376) */
static void record_edge(search_t* s, uint32_t src, uint32_t dst, int dir)
{
    if (dir == FWD)
        hist_add_edge(&s->vt, FWD, dst, src);
    else
        hist_add_edge(&s->vt, REV, src, dst);
}

/* BidirectionalBFS.h:14-76 for one edge (src -> dst in graph orientation) */
static int visit_edge(search_t* s, uint32_t src, uint32_t dst, int dir, u32vec* q, uint32_t qhead)
{
    vtab_t* t = &s->vt;
    uint32_t v = dir == FWD ? dst : src;
    uint32_t parent = dir == FWD ? src : dst;
    if (t->vt[v].color[!dir] != 0) {
        /* common edge */
        if (bump_cost(s) == V_ABORT)
            return V_ABORT;
        if (t->vt[parent].depth[dir] >= s->max_depth[dir])
            return V_SKIP;
        int dup = 0;
        for (uint32_t i = 0; i < s->n_ce; i++)
            if (s->ce_u[i] == src && s->ce_v[i] == dst)
                dup = 1;
        if (!dup && s->n_ce < 64) {
            s->ce_u[s->n_ce] = src;
            s->ce_v[s->n_ce] = dst;
            s->n_ce++;
        }
        if (s->max_paths != NO_LIMIT && s->n_ce > s->max_paths) {
            s->too_many_paths = 1;
            return V_ABORT;
        }
        record_edge(s, src, dst, FWD);
        record_edge(s, src, dst, REV);
        return V_SUCCESS;
    }
    if (t->vt[v].color[dir] == 0) {
        /* discover_vertex with the current queue length */
        uint32_t active = q->n - qhead;
        if (s->max_branches != NO_LIMIT && active >= s->max_branches) {
            s->too_many_branches = 1;
            return V_ABORT;
        }
        /* tree_edge */
        if (bump_cost(s) == V_ABORT)
            return V_ABORT;
        uint16_t pd = t->vt[parent].depth[dir];
        if (pd == s->max_depth[dir])
            return V_SKIP;
        t->vt[v].depth[dir] = (uint16_t)(pd + 1);
        record_edge(s, src, dst, dir);
        t->vt[v].color[dir] = 1;
        u32_push(q, v);
        return V_SUCCESS;
    }
    /* non-tree edge */
    if (bump_cost(s) == V_ABORT)
        return V_ABORT;
    record_edge(s, src, dst, dir);
    return V_SUCCESS;
}

/* connectPairs' search part; returns a PathSearchResult and fills s->found */
static int bidi_search(search_t* s)
{
    vtab_t* t = &s->vt;
    const unsigned k = s->g->k;
    u32vec q[2] = { { 0, 0, 0 }, { 0, 0, 0 } };
    uint32_t qh[2] = { 0, 0 };
    char nb[256];

    t->vt[s->start].color[FWD] = 1;
    t->vt[s->goal].color[REV] = 1;
    /* discover_vertex for both origins: queue sizes are 0, so only the
     * max_branches == 0 corner could abort; GoldPolish passes 1000 */
    u32_push(&q[FWD], s->start);
    u32_push(&q[REV], s->goal);

    int dir = FWD, aborted = 0;
    while (!aborted && (qh[FWD] < q[FWD].n || qh[REV] < q[REV].n)) {
        uint32_t u = q[dir].v[qh[dir]++];
        const char* uk = vtab_kmer(t, u);
        for (int b = 0; b < 4 && !aborted; b++) {
            if (dir == FWD) {
                memcpy(nb, uk + 1, k - 1);
                nb[k - 1] = "ACGT"[b];
            } else {
                memcpy(nb + 1, uk, k - 1);
                nb[0] = "ACGT"[b];
            }
            if (!kmer_in_graph(s->g, nb))
                continue;
            uint32_t w = vtab_get(t, nb);
            uk = vtab_kmer(t, u); /* arena may have moved */
            int r = dir == FWD ? visit_edge(s, u, w, FWD, &q[FWD], qh[FWD])
                               : visit_edge(s, w, u, REV, &q[REV], qh[REV]);
            if (r == V_ABORT)
                aborted = 1;
        }
        if (aborted)
            break;
        t->vt[u].color[dir] = 2;
        if (dir == REV && qh[FWD] < q[FWD].n)
            dir = FWD;
        else if (dir == FWD && qh[REV] < q[REV].n)
            dir = REV;
    }
    free(q[0].v);
    free(q[1].v);

    /* pathsToGoal (:236-253) */
    if (s->too_many_paths)
        return R_TOO_MANY_PATHS;
    if (s->too_many_branches)
        return R_TOO_MANY_BRANCHES;
    if (s->max_cost_exceeded)
        return R_MAX_COST;
    int overall = s->found.n ? R_FOUND : R_NO_PATH;
    for (uint32_t i = 0; i < s->n_ce; i++) {
        int rc = build_paths_edge(s, s->ce_u[i], s->ce_v[i]);
        if (rc == R_FOUND)
            overall = R_FOUND;
        else if (rc != R_NO_PATH) {
            overall = rc;
            break;
        }
    }
    return overall;
}

/* --------------------------------------------------------- NW consensus */
static unsigned amb_mask(char c)
{
    static const unsigned tab[26] = { 0x1, 0xe, 0x2, 0xd, 0, 0, 0x4, 0xb, 0, 0, 0xc, 0, 0x3,
                                      0xf, 0, 0, 0, 0x5, 0x6, 0x8, 0, 0x7, 0x9, 0, 0xa, 0 };
    unsigned i = (unsigned)(toupper((unsigned char)c) - 'A');
    return i < 26 ? tab[i] : 0;
}

static char amb_or(char a, char b)
{
    static const char back[17] = "NACMGRSVTWYHKDBN";
    char c = back[(amb_mask(a) | amb_mask(b)) & 15];
    return (islower((unsigned char)a) || islower((unsigned char)b)) ? (char)tolower(c) : c;
}

static int nw_score(char a, char b, char* c)
{
    if (a == b) {
        *c = a;
        return 5;
    }
    *c = amb_or(a, b);
    return (*c == a || *c == b) ? 5 : -4;
}

/* alignGlobal (Align/alignGlobal.cc:144-198) + backtrack (:55-136); returns the
 * consensus (malloc'd), *matches = number of matching columns */
static char* nw_consensus(const char* A, size_t la, const char* B, size_t lb, unsigned* matches)
{
    const int OPEN = -12, EXT = -4, NEG = INT_MIN / 2;
    size_t W = lb + 1;
    int* f = (int*)malloc((la + 1) * W * sizeof(int));
    int* g = (int*)malloc((la + 1) * W * sizeof(int));
    int* h = (int*)malloc((la + 1) * W * sizeof(int));
#define F(i, j) f[(i) * W + (j)]
#define G(i, j) g[(i) * W + (j)]
#define H(i, j) h[(i) * W + (j)]
    for (size_t i = 0; i <= la; i++) {
        F(i, 0) = G(i, 0) = i == 0 ? 0 : OPEN + EXT * ((int)i - 1);
        H(i, 0) = NEG;
    }
    for (size_t j = 0; j <= lb; j++) {
        F(0, j) = H(0, j) = j == 0 ? 0 : OPEN + EXT * ((int)j - 1);
        G(0, j) = NEG;
    }
    char c;
    for (size_t i = 1; i <= la; i++)
        for (size_t j = 1; j <= lb; j++) {
            int a1 = F(i - 1, j) + OPEN, a2 = G(i - 1, j) + EXT;
            G(i, j) = a1 > a2 ? a1 : a2;
            int b1 = F(i, j - 1) + OPEN, b2 = H(i, j - 1) + EXT;
            H(i, j) = b1 > b2 ? b1 : b2;
            int d = F(i - 1, j - 1) + nw_score(A[i - 1], B[j - 1], &c);
            int m = G(i, j) > H(i, j) ? G(i, j) : H(i, j);
            F(i, j) = d > m ? d : m;
        }
    char* cons = (char*)malloc(la + lb + 2);
    size_t n = 0, i = la, j = lb;
    unsigned nm = 0;
    while (i > 0 && j > 0) {
        int fij = F(i, j);
        char a = A[i - 1], b = B[j - 1];
        int sc = nw_score(a, b, &c);
        if (fij == F(i - 1, j - 1) + sc) {
            cons[n++] = c;
            if (sc == 5)
                nm++;
            i--;
            j--;
        } else if (fij == F(i - 1, j) + OPEN || fij == G(i - 1, j) + EXT) {
            while (G(i, j) == G(i - 1, j) + EXT) {
                cons[n++] = (char)tolower((unsigned char)A[i - 1]);
                i--;
            }
            cons[n++] = (char)tolower((unsigned char)A[i - 1]);
            i--;
        } else {
            while (H(i, j) == H(i, j - 1) + EXT) {
                cons[n++] = (char)tolower((unsigned char)B[j - 1]);
                j--;
            }
            cons[n++] = (char)tolower((unsigned char)B[j - 1]);
            j--;
        }
    }
    while (i > 0)
        cons[n++] = (char)tolower((unsigned char)A[--i]);
    while (j > 0)
        cons[n++] = (char)tolower((unsigned char)B[--j]);
    for (size_t a = 0, b = n; a + 1 < b; a++, b--) {
        char tmp = cons[a];
        cons[a] = cons[b - 1];
        cons[b - 1] = tmp;
    }
    cons[n] = 0;
#undef F
#undef G
#undef H
    free(f);
    free(g);
    free(h);
    *matches = nm;
    return cons;
}

/* exported for the known-answer tests (tests/golden/nw_kat.json, made from the reference's own
 * alignGlobal.cc + Sequence.cpp by oracle/_ref/align_ref) */
size_t gpo_nw_consensus(const char* a, size_t la, const char* b, size_t lb, char* out, unsigned* matches)
{
    char* c = nw_consensus(a, la, b, lb, matches);
    size_t n = strlen(c);
    memcpy(out, c, n + 1);
    free(c);
    return n;
}

/* align() of Align/alignGlobal.h:73-83: alignPair for two sequences, alignMulti's left fold for
 * more (whose `matches = min(0, ...)` is always 0, :64-67) -- what connect_pairs() below does with
 * the connecting sequences.  out needs sum(lens) + 1 bytes. */
size_t gpo_align_multi(unsigned n, const char* const* seqs, const size_t* lens, char* out, unsigned* matches)
{
    unsigned m = 0;
    char* cons = nw_consensus(seqs[0], lens[0], seqs[1], lens[1], &m);
    for (unsigned i = 2; i < n; i++) {
        char* nx = nw_consensus(cons, strlen(cons), seqs[i], lens[i], &m);
        free(cons);
        cons = nx;
    }
    if (n > 2)
        m = 0;
    size_t l = strlen(cons);
    memcpy(out, cons, l + 1);
    free(cons);
    *matches = m;
    return l;
}

void gpo_revcomp(const char* s, size_t n, char* out) { revcomp(s, n, out); }
char gpo_amb_or(char a, char b) { return amb_or(a, b); }

/* ----------------------------------------------------------- connectPairs */
/* read1 = upper(left flank), read2 = revcomp(upper(right flank)).  On success
 * returns the merged sequence (malloc'd), else NULL; *why = PathSearchResult or
 * -1 for "no start/goal k-mer". */
static char* connect_pairs(graph_t* g, const char* r1, size_t l1, const char* r2, size_t l2,
                           const gpo_sealer_params* p, int* why, unsigned* n_paths)
{
    *n_paths = 0;
    const unsigned k = g->k;
    *why = R_NO_PATH;
    if (l1 < k || l2 < k)
        return NULL; /* counted by merge() as noStartOrGoalKmer (found* stay false) */
    unsigned sp = start_kmer_pos(g, r1, l1, 3);
    unsigned gp = start_kmer_pos(g, r2, l2, 3);
    if (sp == NO_MATCH || gp == NO_MATCH) {
        *why = -1;
        return NULL;
    }
    search_t S;
    memset(&S, 0, sizeof S);
    S.g = g;
    vtab_init(&S.vt, k);
    char gk[256];
    revcomp(r2 + gp, k, gk);
    S.start = vtab_get(&S.vt, r1 + sp);
    S.goal = vtab_get(&S.vt, gk);
    unsigned max_merged = p->max_gap_len + 2 * p->flank_len;
    unsigned max_path = max_merged - k + 1 - sp - gp;
    int min_i = (int)(0u - k + 1 - sp - gp);
    unsigned min_path = (unsigned)(min_i > 0 ? min_i : 0);
    size_t need1 = l1 - k + 1 - sp, need2 = l2 - k + 1 - gp;
    unsigned need = (unsigned)(need1 > need2 ? need1 : need2);
    if (need > min_path)
        min_path = need;
    S.max_paths = p->max_paths;
    S.max_branches = p->max_branches;
    S.max_cost = p->max_cost;
    S.min_len = (uint16_t)min_path;
    S.max_len = (uint16_t)max_path;
    uint16_t md = (uint16_t)(S.max_len - 1);
    S.max_depth[FWD] = (uint16_t)(md / 2 + md % 2);
    S.max_depth[REV] = (uint16_t)(md / 2);
    if (S.start == S.goal && 1 >= S.min_len)
        pl_push(&S.found, &S.start, 1);

    int rc = bidi_search(&S);
    *why = rc;
    char* merged = NULL;
    if (rc == R_FOUND) {
        /* connecting sequences */
        uint32_t np = S.found.n;
        *n_paths = np;
        char** conn = (char**)malloc(np * sizeof(char*));
        for (uint32_t i = 0; i < np; i++) {
            const path_t* P = &S.found.p[i];
            conn[i] = (char*)malloc(k + P->n + 1);
            memcpy(conn[i], vtab_kmer(&S.vt, P->v[0]), k);
            for (uint32_t j = 1; j < P->n; j++)
                conn[i][k + j - 1] = vtab_kmer(&S.vt, P->v[j])[k - 1];
            conn[i][k + P->n - 1] = 0;
        }
        char* cons;
        if (np == 1) {
            cons = conn[0];
        } else {
            /* alignPair for 2, alignMulti (left fold) for more */
            unsigned m;
            cons = nw_consensus(conn[0], strlen(conn[0]), conn[1], strlen(conn[1]), &m);
            for (uint32_t i = 2; i < np; i++) {
                char* nx = nw_consensus(cons, strlen(cons), conn[i], strlen(conn[i]), &m);
                free(cons);
                cons = nx;
            }
        }
        size_t lc = strlen(cons);
        merged = (char*)malloc(sp + lc + gp + 1);
        memcpy(merged, r1, sp);
        memcpy(merged + sp, cons, lc);
        char* suf = (char*)malloc(gp + 1);
        revcomp(r2, gp, suf);
        memcpy(merged + sp + lc, suf, gp);
        merged[sp + lc + gp] = 0;
        free(suf);
        /* maskNew(read1, read2, merged, mask = 0): upper-cases positions that
         * agree with the flanks (DBGBloomAlgorithms.h:183-210) */
        size_t lm = sp + lc + gp;
        for (size_t i = 0; i < l1 && i < lm; i++)
            if (r1[i] == merged[i])
                merged[i] = (char)toupper((unsigned char)r1[i]);
        if (l2 <= lm) {
            char* r2rc = (char*)malloc(l2 + 1);
            revcomp(r2, l2, r2rc);
            for (size_t i = 0; i < l2; i++) {
                size_t at = i + lm - l2;
                if (r2rc[i] == merged[at])
                    merged[at] = (char)toupper((unsigned char)r2rc[i]);
            }
            free(r2rc);
        }
        if (np > 1)
            free(cons);
        for (uint32_t i = 0; i < np; i++)
            free(conn[i]);
        free(conn);
    }
    pl_free(&S.found);
    vtab_free(&S.vt);
    return merged;
}

/* ------------------------------------------------------------- the record */
typedef struct {
    size_t ls, le, rs, re; /* left [ls,le)  right [rs,re) */
    char* r1;
    char* r2;
    size_t l1, l2;
    int alive;  /* still in `flanks` */
    char* fill; /* merged sequence once closed */
} gap_t;

void gpo_sealer_default_params(gpo_sealer_params* p, unsigned first_k)
{
    /* scripts/goldpolish-make:69-72 and sealer.cpp:131-224 defaults */
    p->flank_len = first_k;
    p->max_gap_len = 800;
    p->max_paths = 10;
    p->max_branches = 1000;
    p->max_cost = 100000;
    p->mem_limit = 500ull * 1024 * 1024;
    p->lower = 1;
}

size_t gpo_sealer_out_bound(size_t len) { return 2 * len + 4096; }

static int same_str(const char* a, size_t la, const char* b, size_t lb)
{
    return la == lb && memcmp(a, b, la) == 0;
}

size_t gpo_sealer_record(const char* seq, size_t len, unsigned nk, const unsigned* k_values,
                         const uint8_t* const* bfs, size_t bf_bytes, unsigned hash_num,
                         const gpo_sealer_params* p, char* out, gpo_sealer_stats* st)
{
    const char* gapset = p->lower ? "Nnatcgurykmswbdhvx" : "Nn";
    gap_t* gaps = NULL;
    size_t ng = 0, capg = 0;

    /* findFlanks (:695-733) */
    size_t offset = 0;
    for (;;) {
        size_t s = offset;
        while (s < len && !strchr(gapset, seq[s]))
            s++;
        if (s >= len)
            break;
        if (st)
            st->gaps_found++;
        size_t e = s;
        while (e < len && strchr(gapset, seq[e]))
            e++;
        if (e >= len)
            break; /* "sequence ends with a gap" warning */
        size_t re = e;
        while (re < len && !strchr(gapset, seq[re]))
            re++;
        gap_t G;
        memset(&G, 0, sizeof G);
        long ls = (long)s - (long)p->flank_len;
        G.ls = (size_t)(ls > (long)offset ? ls : (long)offset);
        G.le = s;
        G.rs = e;
        G.re = re < e + p->flank_len ? re : e + p->flank_len;
        G.l1 = G.le - G.ls;
        G.l2 = G.re - G.rs;
        G.r1 = (char*)malloc(G.l1 + 1);
        for (size_t i = 0; i < G.l1; i++)
            G.r1[i] = (char)toupper((unsigned char)seq[G.ls + i]);
        G.r1[G.l1] = 0;
        char* rf = (char*)malloc(G.l2 + 1);
        for (size_t i = 0; i < G.l2; i++)
            rf[i] = (char)toupper((unsigned char)seq[G.rs + i]);
        G.r2 = (char*)malloc(G.l2 + 1);
        revcomp(rf, G.l2, G.r2);
        free(rf);
        G.alive = 1;
        /* flanks[read1][read2] = gap: an identical flank pair overwrites (:729) */
        for (size_t i = 0; i < ng; i++)
            if (gaps[i].alive && same_str(gaps[i].r1, gaps[i].l1, G.r1, G.l1) &&
                same_str(gaps[i].r2, gaps[i].l2, G.r2, G.l2))
                gaps[i].alive = 0;
        if (ng == capg) {
            capg = capg ? capg * 2 : 8;
            gaps = (gap_t*)realloc(gaps, capg * sizeof(gap_t));
        }
        gaps[ng++] = G;
        offset = e;
    }

    /* k sweep (:1029-1065) */
    for (unsigned ki = 0; ki < nk; ki++) {
        graph_t g = { bfs[ki], bf_bytes, hash_num, k_values[ki], 0 };
        uint8_t* closed_now = (uint8_t*)calloc(ng ? ng : 1, 1);
        for (size_t i = 0; i < ng; i++) {
            if (!gaps[i].alive)
                continue;
            int why;
            unsigned n_paths;
            char* m = connect_pairs(&g, gaps[i].r1, gaps[i].l1, gaps[i].r2, gaps[i].l2, p, &why,
                                    &n_paths);
            if (st && n_paths == 1)
                st->unique_path++;
            if (st && n_paths > 1)
                st->multiple_paths++;
            if (st) {
                if (why == -1 || (why == R_NO_PATH && (gaps[i].l1 < g.k || gaps[i].l2 < g.k)))
                    st->no_start_goal++;
                else if (why == R_NO_PATH)
                    st->no_path++;
                else if (why == R_TOO_MANY_PATHS)
                    st->too_many_paths++;
                else if (why == R_TOO_MANY_BRANCHES)
                    st->too_many_branches++;
                else if (why == R_MAX_COST)
                    st->max_cost_exceeded++;
            }
            if (m && m[0]) {
                free(gaps[i].fill);
                gaps[i].fill = m;
                closed_now[i] = 1;
                if (st)
                    st->gaps_closed++;
            } else {
                free(m);
            }
        }
        /* erase every flank group (same left flank) that closed something (:644-654) */
        for (size_t i = 0; i < ng; i++)
            if (closed_now[i])
                for (size_t j = 0; j < ng; j++)
                    if (gaps[j].alive && same_str(gaps[j].r1, gaps[j].l1, gaps[i].r1, gaps[i].l1))
                        gaps[j].alive = 0;
        free(closed_now);
        if (st)
            st->vertex_tests += g.tests;
    }

    /* insertIntoScaffold (:525-558): std::string::replace, right to left */
    size_t cap = gpo_sealer_out_bound(len), n = len;
    memcpy(out, seq, len);
    for (size_t ii = ng; ii-- > 0;) {
        if (!gaps[ii].fill)
            continue;
        size_t pos = gaps[ii].ls, cnt = gaps[ii].re - gaps[ii].ls, fl = strlen(gaps[ii].fill);
        if (pos > n)
            continue; /* std::out_of_range in the reference */
        if (cnt > n - pos)
            cnt = n - pos;
        if (n - cnt + fl > cap)
            continue;
        memmove(out + pos + fl, out + pos + cnt, n - pos - cnt);
        memcpy(out + pos, gaps[ii].fill, fl);
        n = n - cnt + fl;
    }
    for (size_t i = 0; i < ng; i++) {
        free(gaps[i].r1);
        free(gaps[i].r2);
        free(gaps[i].fill);
    }
    free(gaps);
    return n;
}
