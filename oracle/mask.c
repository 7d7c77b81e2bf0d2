/*
 * oracle/mask.c -- CPU ORACLE (test infrastructure, never shipped).
 * The two string transforms between stage B and stage C and after stage C:
 *   /root/reference/scripts/goldpolish-mask:44-72   (extend_gaps, soft mask)
 *   /root/reference/scripts/goldpolish-to-upper:17-21
 * Restated from the Python text; vectors in tests/golden/mask_cases.json are
 * hand-derived from that text (the scripts import btllib, absent here).
 */
#include "gpo.h"

#include <ctype.h>
#include <string.h>

/* character classes of the regex ([ACTG]+|[Nn]+|[actgUNMRWSYKVHDBunmrwsykvhdb]+) */
static int cls_upper_acgt(unsigned char c) { return c == 'A' || c == 'C' || c == 'T' || c == 'G'; }
static int cls_n(unsigned char c) { return c == 'N' || c == 'n'; }
static int cls_other(unsigned char c)
{
    return c != 0 && strchr("actgUNMRWSYKVHDBunmrwsykvhdb", c) != NULL;
}

size_t gpo_mask(const char* in, size_t len, unsigned k, char* out)
{
    /* 1. upper-case the first and last k bases (whole sequence if len < 2k) */
    char* seq = out; /* work in place in the output buffer */
    memcpy(seq, in, len);
    if (len < 2 * (size_t)k) {
        for (size_t i = 0; i < len; i++)
            seq[i] = (char)toupper((unsigned char)seq[i]);
    } else {
        for (size_t i = 0; i < k; i++)
            seq[i] = (char)toupper((unsigned char)seq[i]);
        for (size_t i = len - k; i < len; i++)
            seq[i] = (char)toupper((unsigned char)seq[i]);
    }
    /* 2. re.findall: leftmost match, alternatives tried in order, bytes that
     *    match no class are skipped (and so vanish from the output) */
    size_t o = 0, i = 0;
    while (i < len) {
        unsigned char c = (unsigned char)seq[i];
        int (*cls)(unsigned char) = cls_upper_acgt(c) ? cls_upper_acgt
                                    : cls_n(c)        ? cls_n
                                    : cls_other(c)    ? cls_other
                                                      : NULL;
        if (!cls) {
            i++;
            continue;
        }
        size_t j = i;
        while (j < len && cls((unsigned char)seq[j]))
            j++;
        size_t n = j - i;
        int lower = (seq[i] != 'N') && n < k;
        /* o <= i always, so this forward copy never overtakes its source */
        for (size_t q = 0; q < n; q++)
            seq[o + q] = lower ? (char)tolower((unsigned char)seq[i + q]) : seq[i + q];
        o += n;
        i = j;
    }
    /* 3. strip("Nn"), empty -> "N" */
    size_t b = 0, e = o;
    while (b < e && cls_n((unsigned char)seq[b]))
        b++;
    while (e > b && cls_n((unsigned char)seq[e - 1]))
        e--;
    if (b == e) {
        out[0] = 'N';
        out[1] = 0;
        return 1;
    }
    memmove(out, seq + b, e - b);
    out[e - b] = 0;
    return e - b;
}

void gpo_upper(char* seq, size_t len)
{
    for (size_t i = 0; i < len; i++)
        seq[i] = (char)toupper((unsigned char)seq[i]);
}
