/*
 * evo_oracle.c -- CPU restatement of millanek/evo's pairwise Fst / dxy hot path.
 * TEST INFRASTRUCTURE ONLY (see evo_oracle.h for the pinning status).
 * Plain C99, single thread, no FMA contraction (build with -ffp-contract=off).
 * File:line citations are relative to /root/reference/.
 */
#define _GNU_SOURCE
#include "evo_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ===================================================================== */
/* small helpers                                                          */
/* ===================================================================== */

/* split() of process_vcf_utils.cpp:104-130: std::getline on a stringstream, so a
 * trailing delimiter does not create a trailing empty field and "" has no fields.
 * Splits in place; returns the number of fields. */
static int split_inplace(char *s, char delim, char ***out, int *cap) {
    int n = 0;
    if (*s == '\0') return 0;
    char *p = s;
    for (;;) {
        if (n == *cap) {
            *cap = *cap ? *cap * 2 : 64;
            *out = (char **)realloc(*out, sizeof(char *) * (size_t)*cap);
        }
        (*out)[n++] = p;
        char *q = strchr(p, delim);
        if (!q) break;
        *q = '\0';
        p = q + 1;
        if (*p == '\0') break; /* trailing delimiter: getline yields no further item */
    }
    return n;
}

typedef struct { double *v; size_t n, cap; } dvec;
static void dv_push(dvec *d, double x) {
    if (d->n == d->cap) {
        d->cap = d->cap ? d->cap * 2 : 1024;
        d->v = (double *)realloc(d->v, d->cap * sizeof(double));
    }
    d->v[d->n++] = x;
}
static void dv_clear(dvec *d) { d->n = 0; }
static void dv_free(dvec *d) { free(d->v); d->v = NULL; d->n = d->cap = 0; }

/* line reader over zlib (plain files pass through gzread transparently), mirroring
 * createReader (process_vcf_utils.cpp:872-886) + std::getline */
typedef struct { gzFile f; char *buf; size_t cap; } lreader;
static int lr_open(lreader *r, const char *path) {
    r->f = gzopen(path, "rb");
    r->buf = NULL; r->cap = 0;
    if (!r->f) return -1;
    gzbuffer(r->f, 1 << 20);
    return 0;
}
static char *lr_getline(lreader *r) {
    size_t len = 0;
    if (!r->buf) { r->cap = 1 << 16; r->buf = (char *)malloc(r->cap); }
    for (;;) {
        if (!gzgets(r->f, r->buf + len, (int)(r->cap - len))) {
            if (len == 0) return NULL;
            break;
        }
        len += strlen(r->buf + len);
        if (len && r->buf[len - 1] == '\n') { r->buf[--len] = '\0'; return r->buf; }
        if (len + 1 >= r->cap) { r->cap *= 2; r->buf = (char *)realloc(r->buf, r->cap); }
        else if (gzeof(r->f)) break;
    }
    r->buf[len] = '\0';
    return r->buf;
}
static void lr_close(lreader *r) { if (r->f) gzclose(r->f); free(r->buf); }

/* ===================================================================== */
/* per-site math                                                          */
/* ===================================================================== */

/* process_vcf_fst.h:19-25.  pow(x,2) is x*x at -O2 (and agrees with libm to the bit
 * for every case tests/test_oracle_vs_ref.py tries). */
double orc_fst_numerator(double p1, double p2, int n1, int n2) {
    double power = (p1 - p2) * (p1 - p2);
    double fraction1 = (p1 * (1 - p1)) / (n1 - 1);
    double fraction2 = (p2 * (1 - p2)) / (n2 - 1);
    double numerator = power - fraction1 - fraction2;
    return numerator;
}

/* process_vcf_fst.h:27-30 */
double orc_fst_denominator(double p1, double p2) {
    return (p1 * (1 - p2)) + (p2 * (1 - p1));
}

/* process_vcf_fst.h:33-37 */
double orc_het_simple(double p) {
    double q = 1 - p;
    return 1 - (p * p + q * q);
}

/* process_vcf_fst.h:40-45 */
double orc_het_nei78(double p, int n) {
    double q = 1 - p;
    double simple = 1 - (p * p + q * q);
    return (n * simple) / (n - 1);
}

/* process_vcf_utils.h:350-356 */
double orc_vector_sum(const double *v, size_t n) {
    double sum = 0;
    for (size_t i = 0; i < n; i++) sum += v[i];
    return sum;
}
/* process_vcf_utils.h:369-376 */
double orc_vector_average(const double *v, size_t n) {
    double sum = 0;
    for (size_t i = 0; i < n; i++) sum += v[i];
    return (double)sum / (double)n;
}
/* process_vcf_utils.h:378-385 */
double orc_vector_average_with_region(const double *v, size_t n, int region) {
    double sum = 0;
    for (size_t i = 0; i < n; i++) sum += v[i];
    return (double)sum / (double)region;
}
/* process_vcf_fst.cpp:354-360 */
double orc_calculate_fst(const double *num, const double *den, size_t n) {
    double na = orc_vector_average(num, n);
    double da = orc_vector_average(den, n);
    double fst = na / da;
    if (fst < 0) fst = 0;
    return fst;
}

/* process_vcf_fst.cpp:297-323, the nine-way if-chain kept literally */
double orc_dxy_literal(const int *a, const int *b, int n1, int n2) {
    int sumKij = 0;
    for (int i = 0; i != n1 / 2; i++) {
        for (int j = 0; j != n2 / 2; j++) {
            if (a[i] == 0 && b[j] == 0) {
            } else if (a[i] == 1 && b[j] == 0) { sumKij += 2;
            } else if (a[i] == 0 && b[j] == 1) { sumKij += 2;
            } else if (a[i] == 1 && b[j] == 1) { sumKij += 2;
            } else if (a[i] == 2 && b[j] == 1) { sumKij += 2;
            } else if (a[i] == 1 && b[j] == 2) { sumKij += 2;
            } else if (a[i] == 2 && b[j] == 2) {
            } else if (a[i] == 2 && b[j] == 0) { sumKij += 4;
            } else if (a[i] == 0 && b[j] == 2) { sumKij += 4;
            }
        }
    }
    return (double)sumKij / (n1 * n2);
}

/* process_vcf_fst.cpp:325-351 for one set */
double orc_pi_literal(const int *hap, int n, int nm) {
    double pi = 0;
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) {
            if (hap[i] == 1 && hap[j] == 0) pi++;
            else if (hap[i] == 0 && hap[j] == 1) pi++;
        }
    pi = (2.0 / (nm * (nm - 1))) * pi;
    return pi;
}

/* sumKij = c1*(n2-c2) + c2*(n1-c1) with the FULL chromosome counts */
double orc_dxy_closed(int c1, int c2, int n1, int n2) {
    int sumKij = c1 * (n2 - c2) + c2 * (n1 - c1);
    return (double)sumKij / (n1 * n2);
}
/* k = c*(n-c); (2.0/(nm*(nm-1))) * k with an int product */
double orc_pi_closed(int c, int n, int nm) {
    double k = (double)(c * (n - c));
    return (2.0 / (nm * (nm - 1))) * k;
}

/* ===================================================================== */
/* genotype counting, string level                                        */
/* ===================================================================== */

static void parse_alt_fst(const char *alt_field, int *n_alt, int *allele_as_missing, int *indel) {
    /* process_vcf_fst.cpp:222-233 */
    char *tmp = strdup(alt_field);
    char **f = NULL; int cap = 0;
    int nf = split_inplace(tmp, ',', &f, &cap);
    *n_alt = nf; *allele_as_missing = -1; *indel = 0;
    for (int i = 0; i < nf; i++) {
        if (strcmp(f[i], "*") == 0) { *n_alt = *n_alt - 1; *allele_as_missing = i + 1; }
        if (strlen(f[i]) > 1) *indel = 1;
    }
    free(f); free(tmp);
}

void orc_fst_variant_counts(const char *alt_field, const char *const *genotypes,
                            const size_t *set1_loci, int len1,
                            const size_t *set2_loci, int len2,
                            orc_setcounts *out,
                            int *set1_indiv, int *set2_indiv, int *set1_hap, int *set2_hap) {
    memset(out, 0, sizeof(*out));
    int n1 = len1 * 2, n2 = len2 * 2;
    if (set1_indiv) memset(set1_indiv, 0, sizeof(int) * (size_t)len1);
    if (set2_indiv) memset(set2_indiv, 0, sizeof(int) * (size_t)len2);
    if (set1_hap) memset(set1_hap, 0, sizeof(int) * (size_t)n1);
    if (set2_hap) memset(set2_hap, 0, sizeof(int) * (size_t)n2);
    out->set1_n_withoutMissing = n1;   /* :219-220 */
    out->set2_n_withoutMissing = n2;
    int alleleAsMissing;
    parse_alt_fst(alt_field, &out->n_alt_alleles, &alleleAsMissing, &out->bIndel);
    out->allele_as_missing = alleleAsMissing;
    out->alt_code = 1;
    if (out->n_alt_alleles == 1) {      /* :235 */
        int alt = 1;
        if (alleleAsMissing == 1) alt = 2;   /* :238 */
        out->alt_code = alt;
        for (int s = 0; s < len1; s++) {     /* :243-262 */
            const char *g = genotypes[set1_loci[s]];
            int v1 = g[0] - '0', v2 = g[2] - '0';
            if (v1 == alt) {
                out->set1Count++;
                if (set1_indiv) set1_indiv[s]++;
                if (set1_hap) set1_hap[2 * s]++;
            } else if (g[0] == '.' || v1 == alleleAsMissing) {
                out->set1_n_withoutMissing--;
            }
            if (v2 == alt) {
                out->set1Count++;
                if (set1_indiv) set1_indiv[s]++;
                if (set1_hap) set1_hap[2 * s + 1]++;
            } else if (g[2] == '.' || v2 == alleleAsMissing) {
                out->set1_n_withoutMissing--;
            }
        }
        for (int s = 0; s < len2; s++) {     /* :263-280 */
            const char *g = genotypes[set2_loci[s]];
            int v1 = g[0] - '0', v2 = g[2] - '0';
            if (v1 == alt) {
                out->set2Count++;
                if (set2_indiv) set2_indiv[s]++;
                if (set2_hap) set2_hap[2 * s]++;
            } else if (g[0] == '.' || v1 == alleleAsMissing) {
                out->set2_n_withoutMissing--;
            }
            if (v2 == alt) {
                out->set2Count++;
                if (set2_indiv) set2_indiv[s]++;
                if (set2_hap) set2_hap[2 * s + 1]++;
            } else if (g[2] == '.' || v2 == alleleAsMissing) {
                out->set2_n_withoutMissing--;
            }
        }
    }
}

int orc_fst_site_included(const orc_setcounts *c) {
    /* process_vcf_fst.cpp:543-544 */
    return c->n_alt_alleles == 1 && c->bIndel == 0 && (c->set1Count > 0 || c->set2Count > 0) &&
           (c->set1Count < c->set1_n_withoutMissing || c->set2Count < c->set2_n_withoutMissing);
}

void orc_general_counts(const char *const *genotypes, int n_samples,
                        const int32_t *sample_pop, int n_pops, const uint8_t *use_pop,
                        int *alt, int *alleles, double *aaf,
                        int *alleles_compl, double *aaf_compl) {
    for (int k = 0; k < n_pops; k++) { alt[k] = 0; alleles[k] = 0; aaf[k] = -1.0; }
    /* getBasicCounts, process_vcf_utils.cpp:62-83 */
    for (int i = 0; i < n_samples; i++) {
        int k = sample_pop[i];
        const char *g = genotypes[i];
        if (g[0] == '1') { if (k >= 0) { alt[k]++; alleles[k]++; } }
        else if (g[0] == '0') { if (k >= 0) alleles[k]++; }
        if (g[2] == '1') { if (k >= 0) { alt[k]++; alleles[k]++; } }
        else if (g[2] == '0') { if (k >= 0) alleles[k]++; }
    }
    /* getSetVariantCountsSimple, :55-59 */
    for (int k = 0; k < n_pops; k++)
        if (alleles[k] > 0) aaf[k] = (double)alt[k] / alleles[k];
    /* getComplementCounts, :85-98 */
    for (int k = 0; k < n_pops; k++) {
        int ca = 0, ct = 0;
        for (int u = 0; u < n_pops; u++) {
            if (!use_pop[u]) continue;
            if (u != k) { ca += alt[u]; ct += alleles[u]; }
        }
        aaf_compl[k] = (double)ca / ct;
        alleles_compl[k] = ct;
    }
}

/* ===================================================================== */
/* text -> codes                                                          */
/* ===================================================================== */

uint32_t orc_fst_site_flags(const char *ref_field, const char *alt_field,
                            int *alt_code, int *allele_as_missing) {
    int n_alt, aam, indel;
    parse_alt_fst(alt_field, &n_alt, &aam, &indel);
    *allele_as_missing = aam;
    *alt_code = (n_alt == 1 && aam == 1) ? 2 : 1;
    uint32_t fl = 0;
    if (strlen(ref_field) > 1) fl |= ORC_SITE_SKIP;      /* fst.cpp:529-532 */
    if (n_alt != 1 || indel) fl |= ORC_SITE_SKIP;        /* fst.cpp:543 */
    return fl;
}

uint8_t orc_fst_allele_code(char ch, int alt_code, int allele_as_missing) {
    int v = ch - '0';
    if (v == alt_code) return ORC_ALT;
    if (ch == '.' || v == allele_as_missing) return ORC_MISSING;
    return ORC_REF;
}

uint32_t orc_general_site_flags(const char *ref_field, const char *alt_field) {
    /* evo_FstAgainstAll.cpp:184-188 */
    if (strlen(ref_field) > 1 || strlen(alt_field) > 1 || strcmp(alt_field, "*") == 0)
        return ORC_SITE_SKIP;
    return 0;
}

uint8_t orc_general_allele_code(char ch) {
    if (ch == '1') return ORC_ALT;
    if (ch == '0') return ORC_REF;
    return ORC_MISSING;
}

/* ===================================================================== */
/* code-level restatements                                                */
/* ===================================================================== */

void orc_counts_from_codes(const uint8_t *codes, int64_t n_sites, int n_samples,
                           const int32_t *sample_pop, int n_pops,
                           int32_t *alt, int32_t *nm) {
    int32_t *n_full = (int32_t *)calloc((size_t)n_pops, sizeof(int32_t));
    for (int i = 0; i < n_samples; i++)
        if (sample_pop[i] >= 0) n_full[sample_pop[i]] += 2;
    for (int64_t s = 0; s < n_sites; s++) {
        const uint8_t *row = codes + (size_t)s * 2 * (size_t)n_samples;
        int32_t *a = alt + (size_t)s * (size_t)n_pops;
        int32_t *m = nm + (size_t)s * (size_t)n_pops;
        for (int k = 0; k < n_pops; k++) { a[k] = 0; m[k] = n_full[k]; }
        for (int i = 0; i < n_samples; i++) {
            int k = sample_pop[i];
            if (k < 0) continue;
            for (int h = 0; h < 2; h++) {
                uint8_t c = row[2 * i + h];
                if (c == ORC_ALT) a[k]++;
                else if (c == ORC_MISSING) m[k]--;
            }
        }
    }
    free(n_full);
}

void orc_pair_site_stats(const int32_t *alt, const int32_t *nm, int64_t n_sites, int n_pops,
                         const int32_t *n_full, const int32_t *pair_i, const int32_t *pair_j,
                         int n_pairs, const uint8_t *site_flags,
                         double *num, double *den, double *dxy, uint8_t *include) {
    for (int64_t s = 0; s < n_sites; s++) {
        const int32_t *a = alt + (size_t)s * (size_t)n_pops;
        const int32_t *m = nm + (size_t)s * (size_t)n_pops;
        int skip = site_flags ? (site_flags[s] & ORC_SITE_SKIP) : 0;
        for (int q = 0; q < n_pairs; q++) {
            int i = pair_i[q], j = pair_j[q];
            int c1 = a[i], c2 = a[j], nm1 = m[i], nm2 = m[j];
            size_t o = (size_t)s * (size_t)n_pairs + (size_t)q;
            int inc = !skip && (c1 > 0 || c2 > 0) && (c1 < nm1 || c2 < nm2);
            if (include) include[o] = (uint8_t)inc;
            /* values are defined for every non-skipped site; callers mask with include */
            double p1 = (double)c1 / nm1;           /* utils.h:272-275 */
            double p2 = (double)c2 / nm2;
            if (num) num[o] = orc_fst_numerator(p1, p2, nm1, nm2);
            if (den) den[o] = orc_fst_denominator(p1, p2);
            if (dxy) dxy[o] = orc_dxy_closed(c1, c2, n_full[i], n_full[j]);
        }
    }
}

void orc_pop_site_div(const int32_t *alt, const int32_t *nm, int64_t n_sites, int n_pops,
                      const int32_t *n_full, double *het, double *het_nei, double *pi) {
    for (int64_t s = 0; s < n_sites; s++)
        for (int k = 0; k < n_pops; k++) {
            size_t o = (size_t)s * (size_t)n_pops + (size_t)k;
            double p = (double)alt[o] / n_full[k];            /* fst.cpp:364-365: FULL n */
            if (het) het[o] = orc_het_simple(p);
            if (het_nei) het_nei[o] = orc_het_nei78(p, n_full[k]);
            if (pi) pi[o] = orc_pi_closed(alt[o], n_full[k], nm[o]);
        }
}

int64_t orc_snp_windows(const double *num, const double *den, const double *dxy,
                        const uint8_t *include, int64_t n_sites, int n_pairs, int pair,
                        int size, int step, int64_t max_windows,
                        int64_t *first_var, int64_t *end_site,
                        double *sum_num, double *sum_den, double *sum_dxy,
                        double *fst, double *mean_dxy) {
    dvec vn = {0}, vd = {0}, vx = {0};
    int64_t counted = 0, nw = 0;
    for (int64_t s = 0; s < n_sites; s++) {
        size_t o = (size_t)s * (size_t)n_pairs + (size_t)pair;
        if (!include[o]) continue;
        counted++;
        dv_push(&vn, num[o]); dv_push(&vd, den[o]); dv_push(&vx, dxy[o]);
        if (counted % step == 0 && counted >= size) {      /* fst.cpp:638 */
            const double *wn = vn.v + vn.n - size, *wd = vd.v + vd.n - size, *wx = vx.v + vx.n - size;
            if (nw < max_windows) {
                if (first_var) first_var[nw] = counted - size + 1;
                if (end_site) end_site[nw] = s;
                if (sum_num) sum_num[nw] = orc_vector_sum(wn, (size_t)size);
                if (sum_den) sum_den[nw] = orc_vector_sum(wd, (size_t)size);
                if (sum_dxy) sum_dxy[nw] = orc_vector_sum(wx, (size_t)size);
                if (fst) fst[nw] = orc_calculate_fst(wn, wd, (size_t)size);
                if (mean_dxy) mean_dxy[nw] = orc_vector_average(wx, (size_t)size);
            }
            nw++;
        }
    }
    dv_free(&vn); dv_free(&vd); dv_free(&vx);
    return nw;
}

void orc_global_site_stats(const int32_t *alt, const int32_t *nm, int64_t n_sites, int n_pops,
                           const int32_t *use_idx, int n_use, const int64_t *set_sizes,
                           const uint8_t *site_flags,
                           uint8_t *kept, double *num, double *den) {
    for (int64_t s = 0; s < n_sites; s++) {
        const int32_t *a = alt + (size_t)s * (size_t)n_pops;
        const int32_t *m = nm + (size_t)s * (size_t)n_pops;
        kept[s] = 0;
        if (site_flags && (site_flags[s] & ORC_SITE_SKIP)) continue;
        int ta = 0, tn = 0;
        for (int u = 0; u < n_use; u++) { ta += a[use_idx[u]]; tn += m[use_idx[u]]; }
        /* evo_FstAgainstAll.cpp:198-203 on populationsToUse[0] */
        {
            int k0 = use_idx[0];
            double aaf0 = m[k0] > 0 ? (double)a[k0] / m[k0] : -1.0;   /* utils.cpp:56-58 */
            double aafc0 = (double)(ta - a[k0]) / (tn - m[k0]);       /* utils.cpp:95 */
            if (aaf0 == 0 && aafc0 == 0) continue;
            if (aaf0 == 1 && aafc0 == 1) continue;
        }
        kept[s] = 1;
        for (int u = 0; u < n_use; u++) {
            int k = use_idx[u];
            double p1 = m[k] > 0 ? (double)a[k] / m[k] : -1.0;
            int n1 = m[k];
            int n2 = tn - m[k];
            double p2 = (double)(ta - a[k]) / n2;
            size_t o = (size_t)s * (size_t)n_use + (size_t)u;
            if (n1 >= set_sizes[u] * 0.8) {                            /* :253 */
                num[o] = orc_fst_numerator(p1, p2, n1, n2);
                den[o] = orc_fst_denominator(p1, p2);
            } else {
                num[o] = 0; den[o] = 0;
            }
        }
    }
}

int64_t orc_global_windows(const uint8_t *kept, const uint8_t *site_flags,
                           const double *num, const double *den,
                           int64_t n_sites, int n_use, int size, int step, int64_t max_rows,
                           int64_t *start_site, int64_t *end_site, double *fst) {
    /* deques of fixed length `size`, zero filled: evo_FstAgainstAll.cpp:119-122 */
    double *dq_n = (double *)calloc((size_t)size * (size_t)n_use, sizeof(double));
    double *dq_d = (double *)calloc((size_t)size * (size_t)n_use, sizeof(double));
    int64_t *dq_c = (int64_t *)malloc(sizeof(int64_t) * (size_t)size);
    for (int i = 0; i < size; i++) dq_c[i] = -1;
    int head = 0; /* index of the oldest element in the rings */
    int64_t used = 0, rows = 0;
    for (int64_t s = 0; s < n_sites; s++) {
        if (site_flags && (site_flags[s] & ORC_SITE_SKIP)) continue;
        used++;                                            /* :189 */
        if (!kept[s]) continue;                            /* :198-203 */
        /* push_back + pop_front == overwrite the oldest slot and advance head */
        dq_c[head] = s;
        for (int u = 0; u < n_use; u++) {
            dq_n[(size_t)head * (size_t)n_use + (size_t)u] = num[(size_t)s * (size_t)n_use + (size_t)u];
            dq_d[(size_t)head * (size_t)n_use + (size_t)u] = den[(size_t)s * (size_t)n_use + (size_t)u];
        }
        head = (head + 1) % size;
        if ((used > size || size == step) && (used % step == 0)) {   /* :238, :276 */
            if (rows < max_rows) {
                if (start_site) start_site[rows] = dq_c[head];
                if (end_site) end_site[rows] = s;
                for (int u = 0; u < n_use; u++) {
                    double sn = 0, sd = 0;
                    for (int t = 0; t < size; t++) {
                        int idx = (head + t) % size;        /* oldest -> newest */
                        sn += dq_n[(size_t)idx * (size_t)n_use + (size_t)u];
                        sd += dq_d[(size_t)idx * (size_t)n_use + (size_t)u];
                    }
                    double an = sn / (double)size, ad = sd / (double)size;
                    double f = an / ad;
                    if (f < 0 || ad == 0) f = 0;             /* :278 */
                    if (fst) fst[(size_t)rows * (size_t)n_use + (size_t)u] = f;
                }
            }
            rows++;
        }
    }
    free(dq_n); free(dq_d); free(dq_c);
    return rows;
}

/* ===================================================================== */
/* text-level driver: evo fst (VCF mode)                                  */
/* ===================================================================== */

static char *path_join(const char *dir, const char *name) {
    size_t n = strlen(dir) + strlen(name) + 2;
    char *p = (char *)malloc(n);
    if (dir[0]) snprintf(p, n, "%s/%s", dir, name); else snprintf(p, n, "%s", name);
    return p;
}

static int cmp_str(const void *a, const void *b) {
    return strcmp(*(const char *const *)a, *(const char *const *)b);
}

/* locateSet, process_vcf_utils.cpp:923-936: first match per name, missing names dropped */
static int locate_set(char **sample_names, int n_names, char **set, int n_set, size_t *locs) {
    int n = 0;
    for (int i = 0; i < n_set; i++) {
        int found = -1;
        for (int j = 0; j < n_names; j++)
            if (strcmp(sample_names[j], set[i]) == 0) { found = j; break; }
        if (found < 0) fprintf(stderr, "Did not find the sample: \"%s\"\n", set[i]);
        else locs[n++] = (size_t)found;
    }
    return n;
}

static char *read_line_dup(FILE *f) {
    char *line = NULL; size_t cap = 0;
    ssize_t n = getline(&line, &cap, f);
    if (n < 0) { free(line); return strdup(""); }
    if (n && line[n - 1] == '\n') line[n - 1] = '\0';
    return line;
}

static char *str3(const char *a, const char *b, const char *c) {
    size_t n = strlen(a) + strlen(b) + strlen(c) + 1;
    char *p = (char *)malloc(n);
    snprintf(p, n, "%s%s%s", a, b, c);
    return p;
}

/* first tab-separated field of s equals scaffold?  (split(windowStartEnd,'\t')[0] == scaffold) */
static int first_field_is(const char *s, const char *scaffold, char **second_field_dup) {
    const char *t = strchr(s, '\t');
    size_t l = t ? (size_t)(t - s) : strlen(s);
    if (second_field_dup) {
        *second_field_dup = NULL;
        if (t) {
            const char *u = strchr(t + 1, '\t');
            size_t l2 = u ? (size_t)(u - t - 1) : strlen(t + 1);
            *second_field_dup = strndup(t + 1, l2);
        }
    }
    return strlen(scaffold) == l && strncmp(s, scaffold, l) == 0;
}

/* --accessibleGenomeBED: BedCoordinateFeatures / AccessibleGenome, process_vcf_annotation_tools.h:340-530 ------------- */

/* getNumBPinRegion (:360-402) on the features of one scaffold, starts / ends in file order: the first feature whose end
 * is >= start by std::lower_bound on the ends (the halving search of libstdc++, so unsorted files behave alike), then
 * the four overlap cases summed until a feature starts behind `end`.  Left coordinates are 0-based (BED), the query
 * 1-based; cases 3 and 6 add one base, cases 4 and 5 do not -- restated as written. */
int orc_accessible_bp(const int *starts, const int *ends, int n, int start, int end) {
    int first = 0, len = n;
    while (len > 0) {
        int half = len >> 1, mid = first + half;
        if (ends[mid] < start) { first = mid + 1; len = len - half - 1; }
        else len = half;
    }
    int numBP = 0;
    for (int index = first; index < n && starts[index] <= end; index++) {
        if (starts[index] < start && ends[index] <= end) numBP = numBP + (ends[index] - start) + 1;
        else if (starts[index] >= start && ends[index] <= end) numBP = numBP + (ends[index] - starts[index]);
        else if (starts[index] >= start && ends[index] > end) numBP = numBP + (end - starts[index]);
        else if (starts[index] < start && ends[index] > end) numBP = numBP + (end - start) + 1;
    }
    return numBP;
}

typedef struct { char *name; int *starts, *ends; int n, cap; } bed_scaffold;
struct orc_bed { bed_scaffold *sc; int n, cap; };

static bed_scaffold *bed_find(const orc_bed *b, const char *name) {
    for (int i = 0; i < b->n; i++) if (strcmp(b->sc[i].name, name) == 0) return &b->sc[i];
    return NULL;
}

/* BedFeatureMap[previousScaffold] = features: an assignment, so a scaffold that comes back later in the file replaces
 * its earlier entry (:479-481, :491-493) */
static void bed_store(orc_bed *b, const char *name, int *starts, int *ends, int n) {
    bed_scaffold *e = bed_find(b, name);
    if (!e) {
        if (b->n == b->cap) { b->cap = b->cap ? 2 * b->cap : 16; b->sc = (bed_scaffold *)realloc(b->sc, sizeof(bed_scaffold) * (size_t)b->cap); }
        e = &b->sc[b->n++];
        e->name = strdup(name); e->starts = NULL; e->ends = NULL;
    } else { free(e->starts); free(e->ends); }
    e->n = e->cap = n;
    e->starts = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1)); e->ends = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    memcpy(e->starts, starts, sizeof(int) * (size_t)n); memcpy(e->ends, ends, sizeof(int) * (size_t)n);
}

/* loadBedFeatureMap (:461-496): tab-separated scaffold, start, end; consecutive lines of one scaffold form its entry */
orc_bed *orc_bed_load(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    orc_bed *b = (orc_bed *)calloc(1, sizeof(orc_bed));
    int *st = NULL, *en = NULL, n = 0, cap = 0;
    char *prev = strdup("");
    char *line = NULL; size_t lcap = 0; ssize_t ln;
    while ((ln = getline(&line, &lcap, f)) >= 0) {
        if (ln && line[ln - 1] == '\n') line[ln - 1] = '\0';
        char **fld = NULL; int fcap = 0;
        int nf = split_inplace(line, '\t', &fld, &fcap);
        if (nf >= 3) {
            if (strcmp(fld[0], prev) != 0 && prev[0] != '\0') { bed_store(b, prev, st, en, n); n = 0; }
            if (n == cap) { cap = cap ? 2 * cap : 1024; st = (int *)realloc(st, sizeof(int) * (size_t)cap); en = (int *)realloc(en, sizeof(int) * (size_t)cap); }
            st[n] = atoi(fld[1]); en[n] = atoi(fld[2]); n++;
            free(prev); prev = strdup(fld[0]);
        }
        free(fld);
    }
    free(line);
    fclose(f);
    bed_store(b, prev, st, en, n);                     /* final scaffold (:491-493) */
    free(prev); free(st); free(en);
    return b;
}

void orc_bed_free(orc_bed *b) {
    if (!b) return;
    for (int i = 0; i < b->n; i++) { free(b->sc[i].name); free(b->sc[i].starts); free(b->sc[i].ends); }
    free(b->sc); free(b);
}

/* AccessibleGenome::getAccessibleBPinRegion (:510-512); a scaffold the file does not name has 0 accessible bases (:398-401) */
int orc_bed_accessible_bp(const orc_bed *b, const char *scaffold, int start, int end) {
    const bed_scaffold *e = bed_find(b, scaffold);
    if (!e) return 0;
    return orc_accessible_bp(e->starts, e->ends, e->n, start, end);
}

static const orc_bed *g_fst_bed = NULL;                /* set by orc_fst_run_bed for the duration of one run */

/* ---- Annotation (process_vcf_annotation_tools.h:177-327, process_vcf_annotation_tools.cpp:84-109), as far as
 * getCategoryOfSNP needs it.  A restatement (its loader calls split() of the unbuildable process_vcf_utils.cpp):
 * parity unpinned.  Kept from the reference: a transcript is the run of consecutive lines with one name in column 5
 * ("X-name" marks a partial transcript, which is dropped -- and a partial line also drops the transcript BEFORE it,
 * because the flag is raised before the line is told apart, :268-270); the last transcript of a scaffold is stored
 * only when another scaffold follows, so the last scaffold of the file is never stored (:282-294); a scaffold that
 * comes back replaces its entry; positions are compared as STRINGS (:196, :205); only the first line of a transcript
 * is tried as an exon (:203-209, the break); the exon lines are looked up by transcript name, so of two transcripts
 * with one name in a scaffold the later one's lines serve both (:100).  The reference leaves the partial flag of the
 * file's first transcript uninitialized unless its first line is partial (:265); it is taken as false here. */
typedef struct { char *name, *start, *end, *exon_start, *exon_end, *strand; int *ex_s, *ex_e, n_ex; } annot_tr;   /* ex_*: every line of the transcript's exon entry, as atoi reads it */
typedef struct { char *name; annot_tr *tr; int n; } annot_scaffold;
struct orc_annot { annot_scaffold *sc; int n, cap; };

static char *annot_field(const char *line, int k) {          /* field k of split(line, '\t'), NULL when absent */
    char *tmp = strdup(line); char **f = NULL; int cap = 0;
    int nf = split_inplace(tmp, '\t', &f, &cap);
    char *r = k < nf ? strdup(f[k]) : NULL;
    free(f); free(tmp);
    return r;
}
/* getTranscriptName (:242-252) */
static char *annot_transcript_name(const char *geneColumn, int *partial) {
    char *tmp = strdup(geneColumn); char **f = NULL; int cap = 0;
    int nf = split_inplace(tmp, '-', &f, &cap);
    char *r;
    if (nf == 2) { *partial = 1; r = strdup(f[1]); }
    else if (nf == 1) r = strdup(geneColumn);
    else r = strdup("");
    free(f); free(tmp);
    return r;
}
typedef struct { char **lines; int n, cap; } annot_lines;       /* one transcript */
typedef struct { annot_lines *t; int n, cap; } annot_trs;       /* the transcripts of the scaffold being read */
static void annot_lines_push(annot_lines *t, const char *l) {
    if (t->n == t->cap) { t->cap = t->cap ? 2 * t->cap : 8; t->lines = (char **)realloc(t->lines, sizeof(char *) * (size_t)t->cap); }
    t->lines[t->n++] = strdup(l);
}
static void annot_trs_push(annot_trs *a, annot_lines *t) {      /* takes the lines over */
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 64; a->t = (annot_lines *)realloc(a->t, sizeof(annot_lines) * (size_t)a->cap); }
    a->t[a->n++] = *t; t->lines = NULL; t->n = t->cap = 0;
}
static void annot_lines_clear(annot_lines *t) { for (int i = 0; i < t->n; i++) free(t->lines[i]); free(t->lines); t->lines = NULL; t->n = t->cap = 0; }
static void annot_scaffold_free(annot_scaffold *e) {
    for (int i = 0; i < e->n; i++) { free(e->tr[i].name); free(e->tr[i].start); free(e->tr[i].end); free(e->tr[i].exon_start); free(e->tr[i].exon_end);
                                     free(e->tr[i].strand); free(e->tr[i].ex_s); free(e->tr[i].ex_e); }
    free(e->tr); free(e->name);
}
/* annotationMap[scaffold] = annotation (:285) followed by annotateGeneStartsEnds (annotation_tools.cpp:84-109) */
static int annot_store(orc_annot *a, const char *scaffold, annot_trs *trs) {
    annot_scaffold e; e.name = strdup(scaffold); e.n = trs->n; e.tr = (annot_tr *)calloc((size_t)trs->n + 1, sizeof(annot_tr));
    int bad = 0;
    for (int i = 0; i < trs->n; i++) {
        const char *first = trs->t[i].lines[0], *last = trs->t[i].lines[trs->t[i].n - 1];
        char *strand = annot_field(first, 3);
        e.tr[i].name = annot_field(first, 4);
        if (strand && strcmp(strand, "+") == 0) { e.tr[i].start = annot_field(first, 1); e.tr[i].end = annot_field(last, 2); }
        else { e.tr[i].end = annot_field(first, 2); e.tr[i].start = annot_field(last, 1); }
        if (!strand || !e.tr[i].name || !e.tr[i].start || !e.tr[i].end) bad = 1;
        e.tr[i].strand = strand;
    }
    for (int i = 0; i < trs->n && !bad; i++) {                    /* exons by name: the last transcript of that name (:100) */
        int k = i;
        for (int j = trs->n - 1; j > i; j--) if (strcmp(e.tr[j].name, e.tr[i].name) == 0) { k = j; break; }
        e.tr[i].exon_start = annot_field(trs->t[k].lines[0], 1); e.tr[i].exon_end = annot_field(trs->t[k].lines[0], 2);
        e.tr[i].n_ex = trs->t[k].n;
        e.tr[i].ex_s = (int *)malloc(sizeof(int) * (size_t)trs->t[k].n); e.tr[i].ex_e = (int *)malloc(sizeof(int) * (size_t)trs->t[k].n);
        for (int x = 0; x < trs->t[k].n; x++) {
            char *a = annot_field(trs->t[k].lines[x], 1), *b = annot_field(trs->t[k].lines[x], 2);
            e.tr[i].ex_s[x] = a ? atoi(a) : 0; e.tr[i].ex_e[x] = b ? atoi(b) : 0;
            if (!a || !b) bad = 1;
            free(a); free(b);
        }
    }
    for (int i = 0; i < trs->n; i++) annot_lines_clear(&trs->t[i]);
    trs->n = 0;
    if (bad) { annot_scaffold_free(&e); return -1; }
    for (int i = 0; i < a->n; i++)
        if (strcmp(a->sc[i].name, scaffold) == 0) { annot_scaffold_free(&a->sc[i]); a->sc[i] = e; return 0; }
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 16; a->sc = (annot_scaffold *)realloc(a->sc, sizeof(annot_scaffold) * (size_t)a->cap); }
    a->sc[a->n++] = e;
    return 0;
}
void orc_annot_free(orc_annot *a) {
    if (!a) return;
    for (int i = 0; i < a->n; i++) annot_scaffold_free(&a->sc[i]);
    free(a->sc); free(a);
}
/* Annotation(geneFile, false): loadAnnotationMap (:255-295) */
orc_annot *orc_annot_load(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    orc_annot *a = (orc_annot *)calloc(1, sizeof(orc_annot));
    annot_trs trs = {0}; annot_lines cur = {0};
    char *line = NULL; size_t lcap = 0; ssize_t ln;
    char *currentScaffold = NULL, *currentTranscriptName = NULL;
    int thisPartial = 0, first = 1, bad = 0;
    while (!bad && (ln = getline(&line, &lcap, f)) >= 0) {
        if (ln && line[ln - 1] == '\n') line[ln - 1] = '\0';
        char *sc = annot_field(line, 0), *gene = annot_field(line, 4);
        if (!sc || !gene) { free(sc); free(gene); bad = 1; break; }       /* the reference would index out of range */
        int linePartial = 0;
        char *name = annot_transcript_name(gene, &linePartial);
        if (first) {                                                      /* :260-265 */
            annot_lines_push(&cur, line);
            currentScaffold = sc; currentTranscriptName = name; thisPartial = linePartial; first = 0;
            free(gene);
            continue;
        }
        if (linePartial) thisPartial = 1;                                 /* :269-270, before the line is told apart */
        if (strcmp(sc, currentScaffold) == 0) {
            if (strcmp(name, currentTranscriptName) == 0) { annot_lines_push(&cur, line); free(name); }
            else {
                if (!thisPartial) annot_trs_push(&trs, &cur); else annot_lines_clear(&cur);
                annot_lines_push(&cur, line);
                free(currentTranscriptName); currentTranscriptName = name; thisPartial = linePartial;
            }
            free(sc);
        } else {
            if (!thisPartial) annot_trs_push(&trs, &cur); else annot_lines_clear(&cur);
            if (annot_store(a, currentScaffold, &trs) != 0) bad = 1;
            annot_lines_push(&cur, line);
            free(currentScaffold); currentScaffold = sc;
            free(currentTranscriptName); currentTranscriptName = name; thisPartial = linePartial;
        }
        free(gene);
    }
    /* end of file: the open transcript and the open scaffold are dropped (:293-294) */
    annot_lines_clear(&cur);
    for (int i = 0; i < trs.n; i++) annot_lines_clear(&trs.t[i]);
    free(trs.t); free(line); free(currentScaffold); free(currentTranscriptName);
    fclose(f);
    if (bad || first) { orc_annot_free(a); return NULL; }
    return a;
}
static int annot_str_ge(const char *x, const char *y) { return strcmp(x, y) >= 0; }   /* std::string >=: bytes, then length */
/* getCategoryOfSNP (:190-214) */
const char *orc_annot_category(const orc_annot *a, const char *scaffold, const char *locus) {
    const char *category = "other non-coding";
    const annot_scaffold *e = NULL;
    for (int i = 0; i < a->n; i++) if (strcmp(a->sc[i].name, scaffold) == 0) e = &a->sc[i];
    if (!e) return category;
    for (int i = 0; i < e->n; i++) {
        if (annot_str_ge(locus, e->tr[i].start) && annot_str_ge(e->tr[i].end, locus)) {
            category = "intron";
            if (annot_str_ge(locus, e->tr[i].exon_start) && annot_str_ge(e->tr[i].exon_end, locus)) return "exon";
        }
    }
    return category;
}
/* Annotation::getSNPgeneDetails (process_vcf_annotation_tools.cpp:39-80): positions as integers here; the first
 * transcript of the scaffold whose promoter (3 kb upstream, by strand) or body holds the locus decides.  *gene = its
 * name ("" when none), returns "promoter", "exon", "intron" or "nonCoding". */
const char *orc_annot_gene_details(const orc_annot *a, const char *scaffold, int locus, const char **gene) {
    *gene = "";
    const annot_scaffold *e = NULL;
    for (int i = 0; i < a->n; i++) if (strcmp(a->sc[i].name, scaffold) == 0) e = &a->sc[i];
    if (!e) return "nonCoding";
    for (int i = 0; i < e->n; i++) {
        const annot_tr *t = &e->tr[i];
        int geneStart = atoi(t->start), geneEnd = atoi(t->end);
        if (strcmp(t->strand, "+") == 0) {
            if (locus >= geneStart - 3000 && locus < geneStart) { *gene = t->name; return "promoter"; }
        } else if (strcmp(t->strand, "-") == 0) {
            if (locus > geneEnd && locus <= geneEnd + 3000) { *gene = t->name; return "promoter"; }
        }
        if (locus >= geneStart && locus <= geneEnd) {
            *gene = t->name;
            for (int x = 0; x < t->n_ex; x++) if (locus >= t->ex_s[x] && locus <= t->ex_e[x]) return "exon";
            return "intron";
        }
    }
    return "nonCoding";
}
static const orc_annot *g_fst_annot = NULL;            /* set by orc_fst_run_opts for the duration of one run */
static FILE *g_fst_annot_out = NULL;
static const char *g_fst_anc_sets = NULL;              /* --ancSets file, likewise */

/* The counts getFourSetVariantCounts (process_vcf_utils.cpp:577-627) takes from one line: genotype characters 0 and 2
 * equal to '1', per set; a sample counts once per character however often its column is listed. */
static int four_set_alt(const char *const *genotypes, int n_samples, const size_t *loci, int n_loci) {
    int alt = 0;
    for (int i = 0; i < n_samples; i++) {
        int in = 0;
        for (int k = 0; k < n_loci; k++) if (loci[k] == (size_t)i) { in = 1; break; }
        if (!in) continue;
        if (genotypes[i][0] == '1') alt++;
        if (genotypes[i][0] != '\0' && genotypes[i][1] != '\0' && genotypes[i][2] == '1') alt++;
    }
    return alt;
}

int orc_fst_run(const char *vcf_path, const char *sets_path, const char *samples_path,
                int windowSize, int windowStep, int physicalWindowSize, double regAbove,
                const char *run_name, const char *out_dir, double *summary) {
    lreader vcf;
    if (lr_open(&vcf, vcf_path) != 0) { fprintf(stderr, "Error: could not open %s\n", vcf_path); return -1; }
    FILE *setsFile = fopen(sets_path, "r");
    if (!setsFile) { lr_close(&vcf); return -2; }
    char name[512];
    /* output files, process_vcf_fst.cpp:422-437 */
    FILE *regionsAboveFstFile = NULL; int inRegAbove = 0;
    if (regAbove > 0) {
        snprintf(name, sizeof name, "%s_w_%d_fst_above%g.txt", run_name, windowSize, regAbove);
        char *p = path_join(out_dir, name); regionsAboveFstFile = fopen(p, "w"); free(p);
    }
    snprintf(name, sizeof name, "%s_w_%d_fst.txt", run_name, windowSize);
    char *p1 = path_join(out_dir, name); FILE *pFst = fopen(p1, "w"); free(p1);
    snprintf(name, sizeof name, "%s_w_%d_fixedWindowStats.txt", run_name, physicalWindowSize);
    char *p2 = path_join(out_dir, name); FILE *fw = fopen(p2, "w"); free(p2);
    snprintf(name, sizeof name, "%s_w_%d_heterozygosity.txt", run_name, windowSize);
    char *p3 = path_join(out_dir, name); FILE *pHet = fopen(p3, "w"); free(p3);
    if (!pFst || !fw || !pHet) { lr_close(&vcf); fclose(setsFile); return -3; }
    fprintf(fw, "scaffold\tStart\tEnd\tFst\tDxy\tSet1_pi\tSet2_pi\tAccessible_bp\tSet1_VariantDensity\tSet2_VariantDensity\n");

    /* sets, :440-446 */
    char *set1String = read_line_dup(setsFile), *set2String = read_line_dup(setsFile);
    fclose(setsFile);
    char **set1 = NULL, **set2 = NULL; int c1cap = 0, c2cap = 0;
    int nset1 = split_inplace(set1String, ',', &set1, &c1cap);
    int nset2 = split_inplace(set2String, ',', &set2, &c2cap);
    qsort(set1, (size_t)nset1, sizeof(char *), cmp_str);
    qsort(set2, (size_t)nset2, sizeof(char *), cmp_str);
    /* --ancSets, :403-413 */
    FILE *ancOut = NULL; char *anc1String = NULL, *anc2String = NULL; char **anc1 = NULL, **anc2 = NULL;
    int nanc1 = 0, nanc2 = 0, lenA1 = 0, lenA2 = 0; size_t *anc1Loci = NULL, *anc2Loci = NULL;
    if (g_fst_anc_sets) {
        FILE *af = fopen(g_fst_anc_sets, "r");
        if (!af) { lr_close(&vcf); return -2; }
        snprintf(name, sizeof name, "%sancestralSNPs_fst.txt", run_name);
        char *pa = path_join(out_dir, name); ancOut = fopen(pa, "w"); free(pa);
        if (!ancOut) { fclose(af); lr_close(&vcf); return -3; }
        fprintf(ancOut, "scaffold\tposition\tAncAllelePopulation\tFst\tancSet1_segregating\tancSet2_segregating\n");
        anc1String = read_line_dup(af); anc2String = read_line_dup(af); fclose(af);
        int c1 = 0, c2 = 0;
        nanc1 = split_inplace(anc1String, ',', &anc1, &c1); nanc2 = split_inplace(anc2String, ',', &anc2, &c2);
        qsort(anc1, (size_t)nanc1, sizeof(char *), cmp_str); qsort(anc2, (size_t)nanc2, sizeof(char *), cmp_str);
        anc1Loci = (size_t *)malloc(sizeof(size_t) * (size_t)(nanc1 + 1)); anc2Loci = (size_t *)malloc(sizeof(size_t) * (size_t)(nanc2 + 1));
    }
    int anc_rc = 0;

    int totalVariantNumber = 0, countedVariantNumber = 0;
    char *windowMiddleVariant = strdup("first\tWindow");   /* :451 */
    char *windowStartEnd = strdup("scaffold_0\t0");        /* :452 */
    int windowStart = 0, windowEnd = 0;
    int fixedWindowStart = 0;
    dvec fwDxy = {0}, fwNum = {0}, fwDen = {0}, fwPi1 = {0}, fwPi2 = {0};
    dvec fstNumerators = {0}, fstDenominators = {0}, DxyVector = {0};
    dvec het1S = {0}, het2S = {0}, het1N = {0}, het2N = {0}, pi1V = {0}, pi2V = {0};
    size_t *set1Loci = (size_t *)malloc(sizeof(size_t) * (size_t)(nset1 + 1));
    size_t *set2Loci = (size_t *)malloc(sizeof(size_t) * (size_t)(nset2 + 1));
    int len1 = 0, len2 = 0; int n1 = 0, n2 = 0;
    int *s1indiv = NULL, *s2indiv = NULL, *s1hap = NULL, *s2hap = NULL;
    char **fields = NULL; int fcap = 0;
    char *line;
    while ((line = lr_getline(&vcf)) != NULL) {
        if (line[0] == '#' && line[1] == '#') {
        } else if (line[0] == '#' && line[1] == 'C') {
            char *hdr = strdup(line);
            char **hf = NULL; int hcap = 0;
            int nh = split_inplace(hdr, '\t', &hf, &hcap);
            char **sampleNames = NULL; int nNames = 0; char *namesBlob = NULL;
            if (!samples_path || !samples_path[0]) {       /* :482-485 */
                nNames = nh - 9; if (nNames < 0) nNames = 0;
                sampleNames = hf + 9;
            } else {                                        /* :486-488 */
                FILE *sf = fopen(samples_path, "r");
                size_t cap = 0; char *l = NULL; ssize_t r; int ncap = 0;
                while (sf && (r = getline(&l, &cap, sf)) >= 0) {
                    if (r && l[r - 1] == '\n') l[r - 1] = '\0';
                    if (nNames == ncap) { ncap = ncap ? ncap * 2 : 64; sampleNames = (char **)realloc(sampleNames, sizeof(char *) * (size_t)ncap); }
                    sampleNames[nNames++] = strdup(l);
                }
                free(l); if (sf) fclose(sf);
                namesBlob = (char *)1;
            }
            len1 = locate_set(sampleNames, nNames, set1, nset1, set1Loci);
            len2 = locate_set(sampleNames, nNames, set2, nset2, set2Loci);
            n1 = len1 * 2; n2 = len2 * 2;                   /* :491 */
            if (ancOut) {                                   /* :497-505 */
                lenA1 = locate_set(sampleNames, nNames, anc1, nanc1, anc1Loci);
                lenA2 = locate_set(sampleNames, nNames, anc2, nanc2, anc2Loci);
            }
            s1indiv = (int *)malloc(sizeof(int) * (size_t)(len1 + 1));
            s2indiv = (int *)malloc(sizeof(int) * (size_t)(len2 + 1));
            s1hap = (int *)malloc(sizeof(int) * (size_t)(n1 + 2));
            s2hap = (int *)malloc(sizeof(int) * (size_t)(n2 + 2));
            if (namesBlob) { for (int i = 0; i < nNames; i++) free(sampleNames[i]); free(sampleNames); }
            free(hf); free(hdr);
            if (windowSize > 0) {                           /* :507-520 */
                if (windowSize == windowStep) {
                    if (windowSize == 1) {
                        fprintf(pFst, "var_num\tscaffold\tPosition\tFst\tDxy_thisVariant\n");
                    } else {
                        fprintf(pHet, "scaffold\tStart\tEnd\tSet1_heterozygosity\tSet2_heterozygosity\tSet1_heterozygosity_Nei\tSet2_heterozygosity_Nei\tSet1_nucleotideDiversity_pi\tSet2_nucleotideDiversity_pi\n");
                        fprintf(pFst, "var_num\tscaffold\tStart\tEnd\tFst\tDxy_onlyVariants\tDxy_AllSites\twindowSize\n");
                    }
                    if (regAbove > 0) fprintf(regionsAboveFstFile, "scaffold\tStart\tEnd\n");
                } else {
                    fprintf(pHet, "Middle_SNP_position\tSet1_heterozygosity\tSet2_heterozygosity\tSet1_heterozygosity_Nei\tSet2_heterozygosity_Nei\tSet1_nucleotideDiversity_pi\tSet2_nucleotideDiversity_pi\n");
                    fprintf(pFst, "var_num\tscaffold\tStart\tEnd\tFst\tDxy_onlyVariants\tDxy_AllSites\twindowSize\n");
                }
            }
        } else {
            totalVariantNumber++;
            int nf = split_inplace(line, '\t', &fields, &fcap);
            if (nf < 9) continue; /* the reference would index out of range */
            const char *scaffold = fields[0];
            const char *loc = fields[1];
            if (strlen(fields[3]) > 1) continue;            /* :529-532 */
            char *infoCopy = ancOut ? strdup(fields[7]) : NULL;   /* :526 */
            orc_setcounts c;
            orc_fst_variant_counts(fields[4], (const char *const *)(fields + 9), set1Loci, len1, set2Loci, len2,
                                   &c, s1indiv, s2indiv, s1hap, s2hap);
            if (orc_fst_site_included(&c)) {
                countedVariantNumber++;
                double p1 = (double)c.set1Count / c.set1_n_withoutMissing;
                double p2 = (double)c.set2Count / c.set2_n_withoutMissing;
                double FstNumerator = orc_fst_numerator(p1, p2, c.set1_n_withoutMissing, c.set2_n_withoutMissing);
                dv_push(&fstNumerators, FstNumerator); dv_push(&fwNum, FstNumerator);
                double FstDenominator = orc_fst_denominator(p1, p2);
                dv_push(&fstDenominators, FstDenominator); dv_push(&fwDen, FstDenominator);
                double thisSNPDxy = orc_dxy_literal(s1indiv, s2indiv, n1, n2);
                dv_push(&DxyVector, thisSNPDxy); dv_push(&fwDxy, thisSNPDxy);
                double pf1 = (double)c.set1Count / n1, pf2 = (double)c.set2Count / n2;   /* :364-365 */
                double h1 = orc_het_simple(pf1), h2 = orc_het_simple(pf2);
                double hn1 = orc_het_nei78(pf1, n1), hn2 = orc_het_nei78(pf2, n2);
                double pi1 = orc_pi_literal(s1hap, n1, c.set1_n_withoutMissing);
                double pi2 = orc_pi_literal(s2hap, n2, c.set2_n_withoutMissing);
                dv_push(&fwPi1, pi1); dv_push(&fwPi2, pi2);
                dv_push(&pi1V, pi1); dv_push(&pi2V, pi2);
                dv_push(&het1S, h1); dv_push(&het2S, h2);
                dv_push(&het1N, hn1); dv_push(&het2N, hn2);
                if (ancOut) {                                   /* :563-588 */
                    double thisSNPFst = FstNumerator / FstDenominator;
                    if (thisSNPFst < 0) thisSNPFst = 0;
                    char **info = NULL, **kv = NULL; int icap = 0, kcap = 0;
                    int ni = split_inplace(infoCopy, ';', &info, &icap);
                    int nk = ni > 0 ? split_inplace(info[ni - 1], '=', &kv, &kcap) : 0;
                    if (nk < 2) anc_rc = -7;                    /* the reference indexes [1] of that split regardless */
                    else {
                        const char *AA = kv[1];
                        const char *const *gt = (const char *const *)(fields + 9);
                        int ns = nf - 9;
                        int a1 = four_set_alt(gt, ns, set1Loci, len1), a2 = four_set_alt(gt, ns, set2Loci, len2);
                        int a3 = four_set_alt(gt, ns, anc1Loci, lenA1), a4 = four_set_alt(gt, ns, anc2Loci, lenA2);
                        double f3, f4;
                        if (strcmp(AA, fields[3]) == 0) {       /* derived = ALT */
                            double d1 = (double)a1 / (len1 * 2), d2 = (double)a2 / (len2 * 2);
                            f3 = (double)a3 / (lenA1 * 2); f4 = (double)a4 / (lenA2 * 2);
                            fprintf(ancOut, "%s\t%s\t%g\t%g\t", scaffold, loc, d1 - d2, thisSNPFst);
                        } else if (strcmp(AA, fields[4]) == 0) { /* derived = REF */
                            double d1 = (double)(len1 * 2 - a1) / (len1 * 2), d2 = (double)(len2 * 2 - a2) / (len2 * 2);
                            f3 = (double)(lenA1 * 2 - a3) / (lenA1 * 2); f4 = (double)(lenA2 * 2 - a4) / (lenA2 * 2);
                            fprintf(ancOut, "%s\t%s\t%g\t%g\t", scaffold, loc, d1 - d2, thisSNPFst);
                        } else {
                            f3 = (double)a3 / (lenA1 * 2); f4 = (double)a4 / (lenA2 * 2);
                            fprintf(ancOut, "%s\t%s\t-888\t%g\t", scaffold, loc, thisSNPFst);
                        }
                        fprintf(ancOut, "%s\t%s\n", (f3 > 0) & (f3 < 1) ? "1" : "0", (f4 > 0) & (f4 < 1) ? "1" : "0");
                    }
                    free(info); free(kv);
                }
                if (g_fst_annot)                                /* :558-562 */
                    fprintf(g_fst_annot_out, "%s\t%g\t%g\t%s\t%s\n", orc_annot_category(g_fst_annot, scaffold, loc),
                            FstNumerator / FstDenominator, thisSNPDxy, scaffold, loc);

                /* physical windows, :589-629 */
                if (first_field_is(windowStartEnd, scaffold, NULL)) {
                    if (atoi(loc) > (fixedWindowStart + physicalWindowSize)) {
                        int acc = physicalWindowSize;
                        if (g_fst_bed) acc = orc_bed_accessible_bp(g_fst_bed, scaffold, fixedWindowStart, fixedWindowStart + physicalWindowSize);   /* :593-595 */
                        double fDxy = orc_vector_average_with_region(fwDxy.v, fwDxy.n, acc);
                        double fFst = orc_calculate_fst(fwNum.v, fwDen.v, fwNum.n);
                        double fPi1 = orc_vector_average_with_region(fwPi1.v, fwPi1.n, acc);
                        double fPi2 = orc_vector_average_with_region(fwPi2.v, fwPi2.n, acc);
                        int z1 = 0, z2 = 0;
                        for (size_t i = 0; i < fwPi1.n; i++) if (fwPi1.v[i] == 0) z1++;
                        for (size_t i = 0; i < fwPi2.n; i++) if (fwPi2.v[i] == 0) z2++;
                        int nv1 = (int)fwPi1.n - z1, nv2 = (int)fwPi2.n - z2;
                        fprintf(fw, "%s\t%d\t%d\t%g\t%g\t%g\t%g\t%d\t%g\t%g\n", scaffold, fixedWindowStart,
                                fixedWindowStart + physicalWindowSize, fFst, fDxy, fPi1, fPi2, acc,
                                (double)nv1 / acc, (double)nv2 / acc);
                        dv_clear(&fwDxy); dv_clear(&fwNum); dv_clear(&fwDen); dv_clear(&fwPi1); dv_clear(&fwPi2);
                        int without = 0;
                        while (atoi(loc) > (fixedWindowStart + physicalWindowSize)) {
                            if (without > 0) {
                                int acc0 = physicalWindowSize;
                                if (g_fst_bed) acc0 = orc_bed_accessible_bp(g_fst_bed, scaffold, fixedWindowStart, fixedWindowStart + physicalWindowSize);   /* :613-616 */
                                fprintf(fw, "%s\t%d\t%d\tNA\t0\t0\t0\t%d\t0\t0\n", scaffold, fixedWindowStart,
                                        fixedWindowStart + physicalWindowSize, acc0);
                            }
                            fixedWindowStart = fixedWindowStart + physicalWindowSize;
                            without++;
                        }
                    }
                } else {
                    fixedWindowStart = 0;
                    dv_clear(&fwDxy); dv_clear(&fwNum); dv_clear(&fwDen); dv_clear(&fwPi1); dv_clear(&fwPi2);
                }

                if (windowSize == 1) {                      /* :633-636 */
                    double Fst = FstNumerator / FstDenominator;
                    if (Fst < 0) Fst = 0;
                    fprintf(pFst, "%d\t%s\t%s\t%g\t%g\n", countedVariantNumber, scaffold, loc, Fst, thisSNPDxy);
                } else if ((windowSize > 0) && (countedVariantNumber % windowStep == 0) && countedVariantNumber >= windowSize) {
                    size_t W = (size_t)windowSize;
                    const double *wn = fstNumerators.v + fstNumerators.n - W;
                    const double *wd = fstDenominators.v + fstDenominators.n - W;
                    double windowFst = orc_calculate_fst(wn, wd, W); if (windowFst < 0) windowFst = 0;
                    const double *wx = DxyVector.v + DxyVector.n - W;
                    double windowDxy = orc_vector_average(wx, W);
                    if (windowSize == windowStep) {
                        char *s1 = NULL;
                        if (first_field_is(windowStartEnd, scaffold, &s1)) {
                            char *t = str3(windowStartEnd, "\t", loc); free(windowStartEnd); windowStartEnd = t;
                            windowEnd = atoi(loc);
                            double dxyInc = orc_vector_average_with_region(wx, W, windowEnd - windowStart);
                            fprintf(pFst, "%d\t%s\t%g\t%g\t%g\t%zu\n", countedVariantNumber - windowSize + 1,
                                    windowStartEnd, windowFst, windowDxy, dxyInc, W);
                            if (regAbove > 0) {
                                if (windowFst >= regAbove && !inRegAbove) {
                                    inRegAbove = 1;
                                    fprintf(regionsAboveFstFile, "%s\t%s\t", scaffold, s1 ? s1 : "");
                                } else if (windowFst < regAbove && inRegAbove) {
                                    inRegAbove = 0;
                                    fprintf(regionsAboveFstFile, "%s\n", s1 ? s1 : "");
                                }
                            }
                        }
                        free(s1);
                    } else {
                        fprintf(pFst, "%d\t%s\t%g\t%g\t%zu\n", countedVariantNumber - windowSize + 1,
                                windowMiddleVariant, windowFst, windowDxy, W);
                    }
                    double wHetS1 = orc_vector_average(het1S.v + het1S.n - W, W);
                    double wHetS2 = orc_vector_average(het2S.v + het2S.n - W, W);
                    double wHetN1 = orc_vector_average(het1N.v + het1N.n - W, W);
                    double wHetN2 = orc_vector_average(het2N.v + het2N.n - W, W);
                    double wPi1 = orc_vector_average_with_region(pi1V.v + pi1V.n - W, W, windowEnd - windowStart);
                    double wPi2 = orc_vector_average_with_region(pi2V.v + pi2V.n - W, W, windowEnd - windowStart);
                    if (windowSize == windowStep) {
                        if (first_field_is(windowStartEnd, scaffold, NULL)) {
                            fprintf(pHet, "%s\t%g\t%g\t%g\t%g\t%g\t%g\n", windowStartEnd, wHetS1, wHetS2, wHetN1, wHetN2, wPi1, wPi2);
                            free(windowStartEnd); windowStartEnd = str3(scaffold, "\t", loc);
                            windowStart = atoi(loc);
                        } else {
                            free(windowStartEnd); windowStartEnd = str3(scaffold, "\t", "0");
                            windowStart = 0;
                        }
                    } else {
                        fprintf(pHet, "%s\t%g\t%g\t%g\t%g\n", windowMiddleVariant, wHetS1, wHetS2, wHetN1, wHetN2);
                        free(windowMiddleVariant); windowMiddleVariant = str3(scaffold, "\t", loc);
                    }
                }
            }
            free(infoCopy);
        }
    }
    double Fst = orc_calculate_fst(fstNumerators.v, fstDenominators.v, fstNumerators.n);   /* :701 */
    double oS1 = orc_vector_average(het1S.v, het1S.n), oS2 = orc_vector_average(het2S.v, het2S.n);
    double oN1 = orc_vector_average(het1N.v, het1N.n), oN2 = orc_vector_average(het2N.v, het2N.n);
    fprintf(pHet, "#Heterozygosities: \tS1:%g\tS2:%g\tNei1:%g\tNei2%g\n", oS1, oS2, oN1, oN2);   /* :709 */
    if (summary) {
        summary[0] = Fst; summary[1] = oS1; summary[2] = oS2; summary[3] = oN1; summary[4] = oN2;
        summary[5] = countedVariantNumber; summary[6] = totalVariantNumber;
    }
    fclose(pFst); fclose(fw); fclose(pHet); if (regionsAboveFstFile) fclose(regionsAboveFstFile);
    lr_close(&vcf);
    dv_free(&fwDxy); dv_free(&fwNum); dv_free(&fwDen); dv_free(&fwPi1); dv_free(&fwPi2);
    dv_free(&fstNumerators); dv_free(&fstDenominators); dv_free(&DxyVector);
    dv_free(&het1S); dv_free(&het2S); dv_free(&het1N); dv_free(&het2N); dv_free(&pi1V); dv_free(&pi2V);
    free(set1Loci); free(set2Loci); free(s1indiv); free(s2indiv); free(s1hap); free(s2hap);
    free(set1); free(set2); free(set1String); free(set2String);
    free(windowMiddleVariant); free(windowStartEnd); free(fields);
    if (ancOut) fclose(ancOut);
    free(anc1); free(anc2); free(anc1String); free(anc2String); free(anc1Loci); free(anc2Loci);
    return anc_rc;
}

/* orc_fst_run with --accessibleGenomeBED=bed_path (process_vcf_fst.cpp:414-419, :593-595, :613-616): the fixed-window
 * file divides by, and reports, the accessible bases of each window instead of its length. */
int orc_fst_run_bed(const char *vcf_path, const char *sets_path, const char *samples_path,
                    int windowSize, int windowStep, int physicalWindowSize, double regAbove,
                    const char *run_name, const char *out_dir, double *summary, const char *bed_path) {
    orc_bed *bed = NULL;
    if (bed_path && bed_path[0]) {
        bed = orc_bed_load(bed_path);
        if (!bed) { fprintf(stderr, "Error: could not open %s\n", bed_path); return -4; }
    }
    g_fst_bed = bed;
    int rc = orc_fst_run(vcf_path, sets_path, samples_path, windowSize, windowStep, physicalWindowSize, regAbove, run_name, out_dir, summary);
    g_fst_bed = NULL;
    orc_bed_free(bed);
    return rc;
}

/* orc_fst_run with --accessibleGenomeBED, --annot=annot_path (process_vcf_fst.cpp:394-402, :558-562: every included
 * site adds a row {category, Fst of the site, dxy, scaffold, position} to <run_name>SNPcategory_fst.txt) and
 * --ancSets=anc_sets_path (:403-413, :497-505, :563-588: two more sample sets; every included site adds a row
 * {scaffold, position, derived-allele frequency difference of the two sets or -888, Fst clamped at 0, whether each of the
 * two ancestral sets segregates} to <run_name>ancestralSNPs_fst.txt; the ancestral allele is the value of the last INFO
 * entry; returns -7 where that entry has no '=', which the reference does not survive). */
int orc_fst_run_opts(const char *vcf_path, const char *sets_path, const char *samples_path,
                     int windowSize, int windowStep, int physicalWindowSize, double regAbove,
                     const char *run_name, const char *out_dir, double *summary, const char *bed_path, const char *annot_path,
                     const char *anc_sets_path) {
    orc_annot *an = NULL;
    g_fst_anc_sets = anc_sets_path && anc_sets_path[0] ? anc_sets_path : NULL;
    if (annot_path && annot_path[0]) {
        an = orc_annot_load(annot_path);
        if (!an) { fprintf(stderr, "Error: could not read %s\n", annot_path); g_fst_anc_sets = NULL; return -5; }
        char name[512];
        snprintf(name, sizeof name, "%sSNPcategory_fst.txt", run_name);      /* :399 */
        char *p = path_join(out_dir, name); g_fst_annot_out = fopen(p, "w"); free(p);
        if (!g_fst_annot_out) { orc_annot_free(an); g_fst_anc_sets = NULL; return -3; }
        fprintf(g_fst_annot_out, "SNPcategory\tthisSNPFst\tthisSNPDxy\tscaffold\tposition\n");
    }
    g_fst_annot = an;
    int rc = orc_fst_run_bed(vcf_path, sets_path, samples_path, windowSize, windowStep, physicalWindowSize, regAbove, run_name, out_dir, summary, bed_path);
    g_fst_annot = NULL; g_fst_anc_sets = NULL;
    if (g_fst_annot_out) { fclose(g_fst_annot_out); g_fst_annot_out = NULL; }
    orc_annot_free(an);
    return rc;
}

/* ===================================================================== */
/* text-level driver: evo FstGlobal                                       */
/* ===================================================================== */

typedef struct { char *name; char **ids; int n_ids, cap; size_t *pos; int n_pos; } pop_t;

static int cmp_pop(const void *a, const void *b) {
    return strcmp(((const pop_t *)a)->name, ((const pop_t *)b)->name);
}

static const char *g_global_annot_path = NULL;         /* set by orc_fstglobal_run_annot for the duration of one run */
/* the same with --annot (evo_FstAgainstAll.cpp:66-72, :107-116, :215-219, :267-275, :283-297): a row per gene in
 * <run>_FstGlobalGenes_<run>_<SIZE>_<STEP>.txt, written when a SNP of another gene arrives -- after that SNP's own
 * values have joined the vectors, so it is averaged into the gene before it and lost to its own (:267-297). */
int orc_fstglobal_run_annot(const char *vcf_path, const char *pops_path, int windowSize, int windowStep, int fixedWindowSize,
                            const char *run_name, const char *out_dir, const char *annot_path) {
    g_global_annot_path = annot_path && annot_path[0] ? annot_path : NULL;
    int rc = orc_fstglobal_run(vcf_path, pops_path, windowSize, windowStep, fixedWindowSize, run_name, out_dir);
    g_global_annot_path = NULL;
    return rc;
}

int orc_fstglobal_run(const char *vcf_path, const char *pops_path,
                      int windowSize, int windowStep, int fixedWindowSize,
                      const char *run_name, const char *out_dir) {
    lreader vcf;
    if (lr_open(&vcf, vcf_path) != 0) { fprintf(stderr, "Error: could not open %s\n", vcf_path); return -1; }
    FILE *sf = fopen(pops_path, "r");
    if (!sf) { lr_close(&vcf); return -2; }
    /* :83-89  popToIDsMap / IDsToPopMap (later duplicate IDs overwrite the pop of the ID) */
    pop_t *pops = NULL; int n_pops = 0, pcap = 0;
    char **id_names = NULL; char **id_pop = NULL; int n_ids = 0, icap = 0;
    {
        char *l = NULL; size_t cap = 0; ssize_t r;
        while ((r = getline(&l, &cap, sf)) >= 0) {
            if (r && l[r - 1] == '\n') l[r - 1] = '\0';
            char *tab = strchr(l, '\t');
            if (!tab) continue; /* reference would read ID_Pop[1] out of range */
            *tab = '\0';
            char *id = l, *pn = tab + 1;
            char *tab2 = strchr(pn, '\t'); if (tab2) *tab2 = '\0';
            int k = -1;
            for (int i = 0; i < n_pops; i++) if (strcmp(pops[i].name, pn) == 0) { k = i; break; }
            if (k < 0) {
                if (n_pops == pcap) { pcap = pcap ? pcap * 2 : 16; pops = (pop_t *)realloc(pops, sizeof(pop_t) * (size_t)pcap); }
                k = n_pops++;
                memset(&pops[k], 0, sizeof(pop_t));
                pops[k].name = strdup(pn);
            }
            pop_t *P = &pops[k];
            if (P->n_ids == P->cap) { P->cap = P->cap ? P->cap * 2 : 16; P->ids = (char **)realloc(P->ids, sizeof(char *) * (size_t)P->cap); }
            P->ids[P->n_ids++] = strdup(id);
            int found = -1;
            for (int i = 0; i < n_ids; i++) if (strcmp(id_names[i], id) == 0) { found = i; break; }
            if (found < 0) {
                if (n_ids == icap) { icap = icap ? icap * 2 : 64; id_names = (char **)realloc(id_names, sizeof(char *) * (size_t)icap); id_pop = (char **)realloc(id_pop, sizeof(char *) * (size_t)icap); }
                id_names[n_ids] = strdup(id); id_pop[n_ids] = strdup(pn); n_ids++;
            } else { free(id_pop[found]); id_pop[found] = strdup(pn); }
        }
        free(l); fclose(sf);
    }
    qsort(pops, (size_t)n_pops, sizeof(pop_t), cmp_pop);    /* std::map order */
    int *use_idx = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1)); int n_use = 0;
    uint8_t *use_pop = (uint8_t *)calloc((size_t)n_pops + 1, 1);
    for (int k = 0; k < n_pops; k++)
        if (strcmp(pops[k].name, "Outgroup") != 0 && strcmp(pops[k].name, "xxx") != 0) { use_idx[n_use++] = k; use_pop[k] = 1; }
    fprintf(stderr, "There are %d populations \n", n_pops);

    char name[512];
    snprintf(name, sizeof name, "%s_FstGlobal_%d_%d.txt", run_name, windowSize, windowStep);
    char *p1 = path_join(out_dir, name); FILE *outFile = fopen(p1, "w"); free(p1);
    snprintf(name, sizeof name, "%s_FstGlobal_FW%d.txt", run_name, fixedWindowSize);
    char *p2 = path_join(out_dir, name); FILE *outFW = fopen(p2, "w"); free(p2);
    if (!outFile || !outFW) { lr_close(&vcf); return -3; }
    fprintf(outFile, "chr\tpos1\tpos2\tFstGlobal");          /* :101-102 */
    for (int u = 0; u < n_use; u++) fprintf(outFile, "\t%s", pops[use_idx[u]].name);
    fprintf(outFile, "\n");
    fprintf(outFW, "chr\tpos1\tpos2\tnSNPs");                /* :104: pop names go to outFile (sic) */
    for (int u = 0; u < n_use; u++) fprintf(outFile, "\t%s", pops[use_idx[u]].name);
    fprintf(outFW, "\n");
    orc_annot *an = NULL; FILE *outGenes = NULL;             /* :66-72, :107-116 */
    dvec *geneN = NULL, *geneD = NULL;                       /* [n_use][3]: exon, intron, promoter */
    char *currentGene = strdup(""), *previousGene = strdup("");
    if (g_global_annot_path) {
        an = orc_annot_load(g_global_annot_path);
        if (!an) { fprintf(stderr, "Error: could not read %s\n", g_global_annot_path); lr_close(&vcf); return -5; }
        snprintf(name, sizeof name, "%s_FstGlobalGenes_%s_%d_%d.txt", run_name, run_name, windowSize, windowStep);
        char *p3 = path_join(out_dir, name); outGenes = fopen(p3, "w"); free(p3);
        if (!outGenes) { lr_close(&vcf); return -3; }
        fprintf(outGenes, "gene\tnumSNPsExons\tnumSNPsIntrons\tnumSNPs3kbPromoter");
        for (int u = 0; u < n_use; u++) fprintf(outGenes, "\t%s_exons\t%s_wIntrons\t%s_promoter", pops[use_idx[u]].name, pops[use_idx[u]].name, pops[use_idx[u]].name);
        fprintf(outGenes, "\n");
        geneN = (dvec *)calloc((size_t)n_use * 3 + 1, sizeof(dvec)); geneD = (dvec *)calloc((size_t)n_use * 3 + 1, sizeof(dvec));
    }

    size_t nu = (size_t)n_use;
    double *dqN = (double *)calloc((size_t)windowSize * (nu + 1), sizeof(double));
    double *dqD = (double *)calloc((size_t)windowSize * (nu + 1), sizeof(double));
    char **coordDeque = (char **)malloc(sizeof(char *) * (size_t)(windowSize > 0 ? windowSize : 1));
    for (int i = 0; i < windowSize; i++) coordDeque[i] = strdup("0");
    int head = 0;
    int currentWindowStart = 0, currentWindowEnd = currentWindowStart + fixedWindowSize;
    int totalVariantNumber = 0, usedVariantNumber = 0;
    int32_t *sample_pop = NULL; int n_samples = 0;
    char **fields = NULL; int fcap = 0;
    int *alt = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1)), *alleles = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1));
    int *allelesC = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1));
    double *aaf = (double *)malloc(sizeof(double) * (size_t)(n_pops + 1)), *aafC = (double *)malloc(sizeof(double) * (size_t)(n_pops + 1));
    char *line;
    int rc = 0;
    while ((line = lr_getline(&vcf)) != NULL) {
        /* :146 strip every '\r' */
        { char *w = line; for (char *r = line; *r; r++) if (*r != '\r') *w++ = *r; *w = '\0'; }
        if (line[0] == '#' && line[1] == '#') continue;
        else if (line[0] == '#' && line[1] == 'C') {
            char *hdr = strdup(line); char **hf = NULL; int hcap = 0;
            int nh = split_inplace(hdr, '\t', &hf, &hcap);
            n_samples = nh - 9; if (n_samples < 0) n_samples = 0;
            sample_pop = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_samples + 1));
            for (int i = 0; i < n_samples; i++) {            /* :153-155 */
                sample_pop[i] = -1;
                for (int j = 0; j < n_ids; j++)
                    if (strcmp(id_names[j], hf[9 + i]) == 0) {
                        for (int k = 0; k < n_pops; k++) if (strcmp(pops[k].name, id_pop[j]) == 0) sample_pop[i] = k;
                        break;
                    }
            }
            for (int k = 0; k < n_pops; k++) {               /* :158-168 */
                pops[k].pos = (size_t *)malloc(sizeof(size_t) * (size_t)(pops[k].n_ids + 1));
                pops[k].n_pos = locate_set(hf + 9, n_samples, pops[k].ids, pops[k].n_ids, pops[k].pos);
                if (pops[k].n_pos == 0) {
                    fprintf(stderr, "Did not find any samples in the VCF for \"%s\"\n", pops[k].name);
                    rc = -4;   /* the reference aborts on assert(!spPos.empty()) */
                }
            }
            free(hf); free(hdr);
            if (rc) break;
        } else {
            totalVariantNumber++;
            int nf = split_inplace(line, '\t', &fields, &fcap);
            if (nf < 9) continue;
            const char *chr = fields[0]; const char *coord = fields[1];
            double coordDouble = strtod(coord, NULL);
            if (orc_general_site_flags(fields[3], fields[4]) & ORC_SITE_SKIP) continue;   /* :185-188 */
            usedVariantNumber++;
            orc_general_counts((const char *const *)(fields + 9), n_samples, sample_pop, n_pops, use_pop,
                               alt, alleles, aaf, allelesC, aafC);
            int k0 = use_idx[0];
            if (aaf[k0] == 0 && aafC[k0] == 0) continue;     /* :198-203 */
            if (aaf[k0] == 1 && aafC[k0] == 1) continue;
            const char *snpGene = "", *snpCategory = "nonCoding";                        /* :215-219 */
            if (an) {
                snpCategory = orc_annot_gene_details(an, chr, atoi(coord), &snpGene);
                if (snpGene[0]) {
                    free(currentGene); currentGene = strdup(snpGene);
                    if (!previousGene[0]) { free(previousGene); previousGene = strdup(currentGene); }
                }
            }

            if (coordDouble > currentWindowEnd || coordDouble < currentWindowStart) {   /* :222-235 */
                fprintf(outFW, "%s\t%d\t%d\t%d", chr, currentWindowStart, currentWindowEnd, 0);
                for (int u = 0; u < n_use; u++) fprintf(outFW, "\t%g", 0.0);
                fprintf(outFW, "\n");
                if (coordDouble > currentWindowEnd) {
                    currentWindowStart = currentWindowStart + fixedWindowSize; currentWindowEnd = currentWindowEnd + fixedWindowSize;
                } else if (coordDouble < currentWindowStart) {
                    currentWindowStart = 0; currentWindowEnd = 0 + fixedWindowSize;
                }
            }
            /* :237 coordDeque.push_back(coord); pop_front() */
            free(coordDeque[head]); coordDeque[head] = strdup(coord);
            int newest = head;
            head = (head + 1) % windowSize;
            int emit = (usedVariantNumber > windowSize || windowSize == windowStep) && (usedVariantNumber % windowStep == 0);
            if (emit) fprintf(outFile, "%s\t%s\t%s\tN", chr, coordDeque[head], coord);
            for (int u = 0; u < n_use; u++) {                /* :244-282 */
                int k = use_idx[u];
                double p1v = aaf[k], p2v = aafC[k]; int n1v = alleles[k], n2v = allelesC[k];
                double FstNum, FstDenom;
                /* c->setSizes[i]: sizes in ALL-pops order indexed by position in populationsToUse */
                size_t setSize_i = (size_t)pops[u].n_pos;
                if (n1v >= setSize_i * 0.8) {
                    FstNum = orc_fst_numerator(p1v, p2v, n1v, n2v);
                    FstDenom = orc_fst_denominator(p1v, p2v);
                } else { FstNum = 0; FstDenom = 0; }
                dqN[(size_t)newest * (nu + 1) + (size_t)u] = FstNum;
                dqD[(size_t)newest * (nu + 1) + (size_t)u] = FstDenom;
                if (an && snpGene[0]) {                          /* :267-275 */
                    int cat = strcmp(snpCategory, "exon") == 0 ? 0 : strcmp(snpCategory, "intron") == 0 ? 1 : strcmp(snpCategory, "promoter") == 0 ? 2 : -1;
                    if (cat >= 0) { dv_push(&geneN[u * 3 + cat], FstNum); dv_push(&geneD[u * 3 + cat], FstDenom); }
                }
                if (emit) {
                    double sn = 0, sd = 0;
                    for (int t = 0; t < windowSize; t++) {
                        int idx = (head + t) % windowSize;
                        sn += dqN[(size_t)idx * (nu + 1) + (size_t)u];
                        sd += dqD[(size_t)idx * (nu + 1) + (size_t)u];
                    }
                    double an = sn / (double)windowSize, ad = sd / (double)windowSize;
                    double Fst = an / ad; if (Fst < 0 || ad == 0) Fst = 0;
                    fprintf(outFile, "\t%g", Fst);
                }
            }
            if (an && previousGene[0] && strcmp(currentGene, previousGene) != 0) {   /* :283-297 */
                int nEx = (int)geneN[0].n, nIn = (int)geneN[1].n, nPr = (int)geneN[2].n;
                fprintf(outGenes, "%s\t%d\t%d\t%d", previousGene, nEx, nIn, nPr);
                for (int u = 0; u < n_use; u++) {
                    double f[3]; int cnt[3] = {nEx, nIn, nPr};
                    for (int c3 = 0; c3 < 3; c3++) {
                        f[c3] = cnt[c3] > 0 ? orc_vector_average(geneN[u * 3 + c3].v, geneN[u * 3 + c3].n) / orc_vector_average(geneD[u * 3 + c3].v, geneD[u * 3 + c3].n) : 0;
                        if (f[c3] < 0) f[c3] = 0;
                        dv_clear(&geneN[u * 3 + c3]); dv_clear(&geneD[u * 3 + c3]);
                    }
                    fprintf(outGenes, "\t%g\t%g\t%g", f[0], f[1], f[2]);
                }
                fprintf(outGenes, "\n");
                free(previousGene); previousGene = strdup(currentGene);
            }
            if (emit) fprintf(outFile, "\n");
        }
    }
    if (outGenes) fclose(outGenes);
    if (geneN) { for (int i = 0; i < n_use * 3; i++) { dv_free(&geneN[i]); dv_free(&geneD[i]); } free(geneN); free(geneD); }
    orc_annot_free(an); free(currentGene); free(previousGene);
    fclose(outFile); fclose(outFW); lr_close(&vcf);
    for (int i = 0; i < windowSize; i++) free(coordDeque[i]);
    free(coordDeque); free(dqN); free(dqD); free(fields); free(sample_pop);
    free(alt); free(alleles); free(allelesC); free(aaf); free(aafC); free(use_idx); free(use_pop);
    for (int k = 0; k < n_pops; k++) {
        for (int i = 0; i < pops[k].n_ids; i++) free(pops[k].ids[i]);
        free(pops[k].ids); free(pops[k].pos); free(pops[k].name);
    }
    free(pops);
    for (int i = 0; i < n_ids; i++) { free(id_names[i]); free(id_pop[i]); }
    free(id_names); free(id_pop);
    return rc;
}

/* ===================================================================== */
/* PBS (evo_PBS.cpp)                                                     */
/* ===================================================================== */
/* calculatePBSfromAFs, evo_PBS.cpp:69-98 (pow(x,2) == x*x, see the header) */
void orc_pbs_from_afs(double p1, double p2, double p3, double n1, double n2, double n3, double *pbs) {
    double Fst12, Fst13, Fst23;
    double power12 = (p1 - p2) * (p1 - p2);
    double power13 = (p1 - p3) * (p1 - p3);
    double power23 = (p2 - p3) * (p2 - p3);
    double fraction1 = (p1 * (1 - p1)) / (n1 - 1);
    double fraction2 = (p2 * (1 - p2)) / (n2 - 1);
    double fraction3 = (p3 * (1 - p3)) / (n3 - 1);
    double numerator12 = power12 - fraction1 - fraction2;
    double numerator13 = power13 - fraction1 - fraction3;
    double numerator23 = power23 - fraction2 - fraction3;
    double denominator12 = (p1 * (1 - p2)) + (p2 * (1 - p1));
    double denominator13 = (p1 * (1 - p3)) + (p3 * (1 - p1));
    double denominator23 = (p2 * (1 - p3)) + (p3 * (1 - p2));
    if ((p1 == 0 && p2 == 0) || (p1 == 1 && p2 == 1)) Fst12 = 0.0; else Fst12 = numerator12 / denominator12;
    if ((p1 == 0 && p3 == 0) || (p1 == 1 && p3 == 1)) Fst13 = 0.0; else Fst13 = numerator13 / denominator13;
    if ((p2 == 0 && p3 == 0) || (p2 == 1 && p3 == 1)) Fst23 = 0.0; else Fst23 = numerator23 / denominator23;
    if (Fst12 < 0) Fst12 = 0;
    if (Fst13 < 0) Fst13 = 0;
    if (Fst23 < 0) Fst23 = 0;
    if (Fst12 == 1) Fst12 = 1 - (Fst12 / n1);
    if (Fst13 == 1) Fst13 = 1 - (Fst13 / n1);      /* :87 divides by p1AlleleCount for 13 and p2AlleleCount for 23 (sic) */
    if (Fst23 == 1) Fst23 = 1 - (Fst23 / n2);
    double T12 = -log(1 - Fst12), T13 = -log(1 - Fst13), T23 = -log(1 - Fst23);
    pbs[0] = (T12 + T13 - T23) / 2.0;
    pbs[1] = (T12 + T23 - T13) / 2.0;
    pbs[2] = (T13 + T23 - T12) / 2.0;
    for (int x = 0; x < 3; x++) if (pbs[x] < 0) pbs[x] = 0;
}

/* per-site rules of PBSmain (evo_PBS.cpp:304-322) on counts: pbs [n_trios][3][n_sites], used [n_trios][n_sites] */
void orc_pbs_site_stats(const int32_t *alt, const int32_t *nm, int64_t n_sites, int n_pops,
                        const int32_t *trios, int n_trios, const uint8_t *site_flags, double *pbs, uint8_t *used) {
    for (int t = 0; t < n_trios; t++)
        for (int64_t s = 0; s < n_sites; s++) {
            double p[3], n[3], out[3] = {0, 0, 0};
            int ok = !(site_flags && (site_flags[s] & ORC_SITE_SKIP));
            for (int x = 0; x < 3; x++) {
                int k = trios[3 * t + x];
                int a = alt[s * n_pops + k], m = nm[s * n_pops + k];
                p[x] = m > 0 ? (double)a / m : -1.0;
                n[x] = (double)m;
                if (p[x] == -1) ok = 0;
            }
            if (ok && p[0] == 0 && p[1] == 0 && p[2] == 0) ok = 0;
            if (ok && p[0] == 1 && p[1] == 1 && p[2] == 1) ok = 0;
            if (ok) orc_pbs_from_afs(p[0], p[1], p[2], n[0], n[1], n[2], out);
            for (int x = 0; x < 3; x++) pbs[((size_t)t * 3 + (size_t)x) * (size_t)n_sites + (size_t)s] = out[x];
            used[(size_t)t * (size_t)n_sites + (size_t)s] = (uint8_t)ok;
        }
}

/* ===================================================================== */
/* ABS (evo_ABS.cpp)                                                     */
/* ===================================================================== */
/* Hudson Fst of one population pair as calculateABSfromAFs forms it (evo_ABS.cpp:63-91): fa, fb are the two
 * variance terms the reference subtracts for this pair (see the note on their naming in orc_abs_from_afs) */
static double abs_pair_fst(double pa, double pb, double fa, double fb, double na, double nb) {
    double power = (pa - pb) * (pa - pb);
    double numerator = power - fa - fb;
    double denominator = (pa * (1 - pb)) + (pb * (1 - pa));
    double Fst;
    if ((pa == 0 && pb == 0) || (pa == 1 && pb == 1)) Fst = 0.0; else Fst = numerator / denominator;
    if (Fst < 0) Fst = 0;
    if (Fst == 1) Fst = 1 - (Fst / ((na + nb) / 2.0));
    return Fst;
}

/* calculateABSfromAFs, evo_ABS.cpp:58-106.  abs_out[0] = ABS, abs_out[1] = ABSalt.
 * Quirk kept (:66): the variable called fractionX holds W's term pW(1-pW)/(nW-1) and fractionW holds X's, so
 * every pair with W subtracts X's term in W's place and every pair with X subtracts W's (WX itself is unaffected). */
void orc_abs_from_afs(double pW, double pX, double pY, double pZ, double nW, double nX, double nY, double nZ, double *abs_out) {
    double fractionX = (pW * (1 - pW)) / (nW - 1), fractionW = (pX * (1 - pX)) / (nX - 1);
    double fractionY = (pY * (1 - pY)) / (nY - 1), fractionZ = (pZ * (1 - pZ)) / (nZ - 1);
    double FstWX = abs_pair_fst(pW, pX, fractionW, fractionX, nW, nX);
    double FstWY = abs_pair_fst(pW, pY, fractionW, fractionY, nW, nY);
    double FstWZ = abs_pair_fst(pW, pZ, fractionW, fractionZ, nW, nZ);
    double FstXY = abs_pair_fst(pX, pY, fractionX, fractionY, nX, nY);
    double FstXZ = abs_pair_fst(pX, pZ, fractionX, fractionZ, nX, nZ);
    double FstYZ = abs_pair_fst(pY, pZ, fractionY, fractionZ, nY, nZ);
    double TWX = -log(1 - FstWX), TWY = -log(1 - FstWY), TWZ = -log(1 - FstWZ);
    double TXY = -log(1 - FstXY), TXZ = -log(1 - FstXZ), TYZ = -log(1 - FstYZ);
    double TWYTXZ = TWY + TXZ, TWZTXY = TWZ + TXY, TWXTYZ = TWX + TYZ;
    double twoMax = (TWYTXZ >= TWZTXY) ? TWYTXZ : TWZTXY;
    double ABSalt = (twoMax - TWXTYZ) / 2.0;
    double threeMax, threeMin;
    if (TWYTXZ >= TWZTXY && TWYTXZ >= TWXTYZ) { threeMax = TWYTXZ; threeMin = (TWZTXY <= TWXTYZ) ? TWZTXY : TWXTYZ; }
    else if (TWZTXY >= TWYTXZ && TWZTXY >= TWXTYZ) { threeMax = TWZTXY; threeMin = (TWYTXZ <= TWXTYZ) ? TWYTXZ : TWXTYZ; }
    else { threeMax = TWXTYZ; threeMin = (TWYTXZ <= TWZTXY) ? TWYTXZ : TWZTXY; }
    abs_out[0] = (threeMax - threeMin) / 2.0;
    abs_out[1] = ABSalt;
}

/* per-site rules of ABSmain (evo_ABS.cpp:243-262) on counts: abs [n_quartets][2][n_sites], used [n_quartets][n_sites] */
void orc_abs_site_stats(const int32_t *alt, const int32_t *nm, int64_t n_sites, int n_pops,
                        const int32_t *quartets, int n_quartets, const uint8_t *site_flags, double *abs_out, uint8_t *used) {
    for (int t = 0; t < n_quartets; t++)
        for (int64_t s = 0; s < n_sites; s++) {
            double p[4], n[4], out[2] = {0, 0};
            int ok = !(site_flags && (site_flags[s] & ORC_SITE_SKIP));
            for (int x = 0; x < 4; x++) {
                int k = quartets[4 * t + x];
                int a = alt[s * n_pops + k], m = nm[s * n_pops + k];
                p[x] = m > 0 ? (double)a / m : -1.0;
                n[x] = (double)m;
                if (p[x] == -1) ok = 0;
            }
            if (ok && p[0] == 0 && p[1] == 0 && p[2] == 0 && p[3] == 0) ok = 0;
            if (ok && p[0] == 1 && p[1] == 1 && p[2] == 1 && p[3] == 1) ok = 0;
            if (ok) orc_abs_from_afs(p[0], p[1], p[2], p[3], n[0], n[1], n[2], n[3], out);
            for (int x = 0; x < 2; x++) abs_out[((size_t)t * 2 + (size_t)x) * (size_t)n_sites + (size_t)s] = out[x];
            used[(size_t)t * (size_t)n_sites + (size_t)s] = (uint8_t)ok;
        }
}

/* `evo PBS [-w SIZE,STEP] [-f FW] [-n RUN] VCF POPS TRIOS` (PBSmain, evo_PBS.cpp:125-363) without --annot, --af
 * and -i (those come in through orc_pbs_run_annot, orc_set_pbs_af and orc_set_allow_indels), and `evo ABS [-w SIZE,STEP] [-n RUN] VCF POPS QUARTETS` (ABSmain, evo_ABS.cpp:109-293) without --annot.
 * The two mains are the same loop over a different statistic (arity 3 with fixed windows, arity 4 without); the
 * citations below are PBSmain's, ABSmain's lines are noted where the behaviour differs.  Output files go to out_dir. */
static const char *g_pbs_annot_path = NULL;            /* set by orc_pbs_run_annot for the duration of one run */
static int g_pbs_af = 0;                               /* --af of PBS (evo_PBS.cpp:264-271), likewise */
static int g_allow_indels = 0;                         /* -i of PBS / DistOutgroups (evo_PBS.cpp:250-252, evo_distanceToOutgroups.cpp:196-198), likewise */
static int branch_run(int arity, const char *vcf_path, const char *pops_path, const char *trios_path,
                      int windowSize, int windowStep, int fixedWindowSize, const char *run_name, const char *out_dir) {
    const int nval = arity == 3 ? 3 : 2;                         /* values per site: PBS x3, or ABS and ABSalt */
    /* PBS --annot (evo_PBS.cpp:130-136, :176-181, :282-286, :333-341, :348-361): a row per gene and trio */
    orc_annot *an = NULL;
    if (arity == 3 && g_pbs_annot_path) {
        an = orc_annot_load(g_pbs_annot_path);
        if (!an) { fprintf(stderr, "Error: could not read %s\n", g_pbs_annot_path); return -7; }
    }
    FILE **outGenes = NULL; dvec *geneRes = NULL;            /* [trio][9]: exon x3, intron x3, promoter x3 */
    char *currentGene = strdup(""), *previousGene = strdup("");
    lreader vcf;
    if (lr_open(&vcf, vcf_path) != 0) { fprintf(stderr, "Error: could not open %s\n", vcf_path); return -1; }
    FILE *sf = fopen(pops_path, "r");
    FILE *tf = fopen(trios_path, "r");
    if (!sf || !tf) { lr_close(&vcf); return -2; }
    pop_t *pops = NULL; int n_pops = 0, pcap = 0;
    char **id_names = NULL; char **id_pop = NULL; int n_ids = 0, icap = 0;
    {
        char *l = NULL; size_t cap = 0; ssize_t r;
        while ((r = getline(&l, &cap, sf)) >= 0) {           /* :148-154 */
            if (r && l[r - 1] == '\n') l[r - 1] = '\0';
            char *tab = strchr(l, '\t');
            if (!tab) continue;
            *tab = '\0';
            char *id = l, *pn = tab + 1;
            char *tab2 = strchr(pn, '\t'); if (tab2) *tab2 = '\0';
            int k = -1;
            for (int i = 0; i < n_pops; i++) if (strcmp(pops[i].name, pn) == 0) { k = i; break; }
            if (k < 0) {
                if (n_pops == pcap) { pcap = pcap ? pcap * 2 : 16; pops = (pop_t *)realloc(pops, sizeof(pop_t) * (size_t)pcap); }
                k = n_pops++;
                memset(&pops[k], 0, sizeof(pop_t));
                pops[k].name = strdup(pn);
            }
            pop_t *P = &pops[k];
            if (P->n_ids == P->cap) { P->cap = P->cap ? P->cap * 2 : 16; P->ids = (char **)realloc(P->ids, sizeof(char *) * (size_t)P->cap); }
            P->ids[P->n_ids++] = strdup(id);
            int found = -1;
            for (int i = 0; i < n_ids; i++) if (strcmp(id_names[i], id) == 0) { found = i; break; }
            if (found < 0) {
                if (n_ids == icap) { icap = icap ? icap * 2 : 64; id_names = (char **)realloc(id_names, sizeof(char *) * (size_t)icap); id_pop = (char **)realloc(id_pop, sizeof(char *) * (size_t)icap); }
                id_names[n_ids] = strdup(id); id_pop[n_ids] = strdup(pn); n_ids++;
            } else { free(id_pop[found]); id_pop[found] = strdup(pn); }
        }
        free(l); fclose(sf);
    }
    qsort(pops, (size_t)n_pops, sizeof(pop_t), cmp_pop);
    fprintf(stderr, "There are %d populations \n", n_pops);
    /* :163-186 trios and their output files */
    int n_trios = 0, tcap = 0; int *trio = NULL; FILE **outW = NULL, **outFW = NULL;
    int rc = 0;
    {
        char *l = NULL; size_t cap = 0; ssize_t r;
        while ((r = getline(&l, &cap, tf)) >= 0) {
            if (r && l[r - 1] == '\n') l[r - 1] = '\0';
            char *nm4[4] = {l, NULL, NULL, NULL};
            int nf = 1;
            for (char *q = l; *q; q++) if (*q == '\t') { *q = '\0'; if (nf < 4) nm4[nf] = q + 1; nf++; }
            if (nf != arity) { rc = -5; break; }                  /* assert(threePops.size() == 3) / fourPops == 4 */
            if (n_trios == tcap) {
                tcap = tcap ? tcap * 2 : 8;
                trio = (int *)realloc(trio, sizeof(int) * 4 * (size_t)tcap);
                outW = (FILE **)realloc(outW, sizeof(FILE *) * (size_t)tcap); outFW = (FILE **)realloc(outFW, sizeof(FILE *) * (size_t)tcap);
                outGenes = (FILE **)realloc(outGenes, sizeof(FILE *) * (size_t)tcap);
            }
            outGenes[n_trios] = NULL;
            for (int x = 0; x < arity; x++) {
                int k = -1;
                for (int i = 0; i < n_pops; i++) if (strcmp(pops[i].name, nm4[x]) == 0) k = i;
                if (k < 0) rc = -6;                               /* setAAFs.at() would throw */
                trio[4 * n_trios + x] = k;
            }
            char name[1024];
            outFW[n_trios] = NULL;
            if (arity == 3) {
                snprintf(name, sizeof name, "%s_%s_%s_PBS_%s_%d_%d.txt", nm4[0], nm4[1], nm4[2], run_name, windowSize, windowStep);
                char *p1 = path_join(out_dir, name); outW[n_trios] = fopen(p1, "w"); free(p1);
                snprintf(name, sizeof name, "%s_%s_%s_PBS_%s_FW%d.txt", nm4[0], nm4[1], nm4[2], run_name, fixedWindowSize);
                char *p2 = path_join(out_dir, name); outFW[n_trios] = fopen(p2, "w"); free(p2);
                if (!outW[n_trios] || !outFW[n_trios]) { rc = -3; n_trios++; break; }
                fprintf(outW[n_trios], "chr\twStart\twEnd\t%s\t%s\t%s\n", nm4[0], nm4[1], nm4[2]);
                fprintf(outFW[n_trios], "chr\twStart\twEnd\t%s\t%s\t%s\tnFwSNPs1\tnFwSNPs2\tnFwSNPs3\n", nm4[0], nm4[1], nm4[2]);
                if (an) {
                    snprintf(name, sizeof name, "%s_%s_%s_PBSGenes_%s_%d_%d.txt", nm4[0], nm4[1], nm4[2], run_name, windowSize, windowStep);
                    char *p3 = path_join(out_dir, name); outGenes[n_trios] = fopen(p3, "w"); free(p3);
                    if (!outGenes[n_trios]) { rc = -3; n_trios++; break; }
                    fprintf(outGenes[n_trios], "gene\tnumSNPsExons\tnumSNPsIntrons\tnumSNPs3kbPromoter\t%s_exons\t%s_exons\t%s_exons\t%s_wIntrons\t%s_wIntrons\t%s_wIntrons\t%s_promoter\t%s_promoter\t%s_promoter\n",
                            nm4[0], nm4[1], nm4[2], nm4[0], nm4[1], nm4[2], nm4[0], nm4[1], nm4[2]);
                }
            } else {                                              /* ABS.cpp:153-154 */
                snprintf(name, sizeof name, "%s_%s_%s_%s_ABS_%s_%d_%d.txt", nm4[0], nm4[1], nm4[2], nm4[3], run_name, windowSize, windowStep);
                char *p1 = path_join(out_dir, name); outW[n_trios] = fopen(p1, "w"); free(p1);
                if (!outW[n_trios]) { rc = -3; n_trios++; break; }
                fprintf(outW[n_trios], "chr\tpos1\tpos2\tABS\tABSalt\n");
            }
            n_trios++;
        }
        free(l); fclose(tf);
    }
    /* :189-196 per trio: three deques of per-site PBS + one of coordinates, zero-initialised; fixed-window vectors */
    size_t W = (size_t)(windowSize > 0 ? windowSize : 1);
    double *dq = (double *)calloc((size_t)(n_trios + 1) * 4 * W, sizeof(double));
    int *head = (int *)calloc((size_t)n_trios + 1, sizeof(int));
    dvec *fw = (dvec *)calloc((size_t)(n_trios + 1) * 3, sizeof(dvec));
    int *usedVars = (int *)calloc((size_t)n_trios + 1, sizeof(int));
    if (an) geneRes = (dvec *)calloc((size_t)n_trios * 9 + 1, sizeof(dvec));
    int currentWindowStart = 0, currentWindowEnd = currentWindowStart + fixedWindowSize;
    uint8_t *use_pop = (uint8_t *)calloc((size_t)n_pops + 1, 1);
    int32_t *sample_pop = NULL; int n_samples = 0;
    char **fields = NULL; int fcap = 0;
    int *alt = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1)), *alleles = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1));
    int *allelesC = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1));
    double *aaf = (double *)malloc(sizeof(double) * (size_t)(n_pops + 1)), *aafC = (double *)malloc(sizeof(double) * (size_t)(n_pops + 1));
    char *line;
    while (!rc && (line = lr_getline(&vcf)) != NULL) {
        { char *w = line; for (char *r = line; *r; r++) if (*r != '\r') *w++ = *r; *w = '\0'; }   /* :215 */
        if (line[0] == '#' && line[1] == '#') continue;
        else if (line[0] == '#' && line[1] == 'C') {
            char *hdr = strdup(line); char **hf = NULL; int hcap = 0;
            int nh = split_inplace(hdr, '\t', &hf, &hcap);
            n_samples = nh - 9; if (n_samples < 0) n_samples = 0;
            sample_pop = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_samples + 1));
            for (int i = 0; i < n_samples; i++) {
                sample_pop[i] = -1;
                for (int j = 0; j < n_ids; j++)
                    if (strcmp(id_names[j], hf[9 + i]) == 0) {
                        for (int k = 0; k < n_pops; k++) if (strcmp(pops[k].name, id_pop[j]) == 0) sample_pop[i] = k;
                        break;
                    }
            }
            for (int k = 0; k < n_pops; k++) {
                pops[k].pos = (size_t *)malloc(sizeof(size_t) * (size_t)(pops[k].n_ids + 1));
                pops[k].n_pos = locate_set(hf + 9, n_samples, pops[k].ids, pops[k].n_ids, pops[k].pos);
                if (pops[k].n_pos == 0) { fprintf(stderr, "Did not find any samples in the VCF for \"%s\"\n", pops[k].name); rc = -4; }
            }
            free(hf); free(hdr);
        } else {
            int nf = split_inplace(line, '\t', &fields, &fcap);
            if (nf < 9) continue;
            const char *chr = fields[0]; const char *coord = fields[1];
            double coordDouble = strtod(coord, NULL);
            /* :248-256 only biallelic SNPs, unless -i */
            if (strcmp(fields[4], "*") == 0 || (!(arity == 3 && g_allow_indels) && (strlen(fields[3]) > 1 || strlen(fields[4]) > 1))) continue;
            orc_general_counts((const char *const *)(fields + 9), n_samples, sample_pop, n_pops, use_pop,
                               alt, alleles, aaf, allelesC, aafC);
            /* :264-271 --af: at every used line the reference opens stripExtension(setsFile) + "_AF.txt" anew (which
             * empties it) on a stream it never closes, flushes or deletes, so the line written stays in a buffer that
             * is lost at exit (a line of more than a stream buffer, 8 KiB in libstdc++, would reach the file in part;
             * thousands of populations).  What a run leaves is the empty file, here made in out_dir. */
            if (arity == 3 && g_pbs_af) {
                const char *slash = strrchr(pops_path, '/');
                char *base = strdup(slash ? slash + 1 : pops_path);
                { char *dot = strrchr(base, '.'); if (dot) *dot = '\0'; }
                char name[768];
                snprintf(name, sizeof name, "%s_AF.txt", base);
                free(base);
                char *op = path_join(out_dir, name); FILE *af = fopen(op, "w"); free(op);
                if (af) fclose(af);
            }
            const char *snpGene = "", *snpCategory = "nonCoding";                                /* :282-286 */
            if (an) {
                snpCategory = orc_annot_gene_details(an, chr, atoi(coord), &snpGene);
                if (snpGene[0]) {
                    free(currentGene); currentGene = strdup(snpGene);
                    if (!previousGene[0]) { free(previousGene); previousGene = strdup(currentGene); }
                }
            }
            if (arity == 3 && (coordDouble > currentWindowEnd || coordDouble < currentWindowStart)) {   /* :288-303; ABS has no fixed windows */
                for (int i = 0; i < n_trios; i++) {
                    dvec *f0 = &fw[3 * i];
                    int n1 = (int)f0[0].n, n2 = (int)f0[1].n, n3 = (int)f0[2].n;
                    double a1 = 0, a2 = 0, a3 = 0;
                    if (n1 > 0) a1 = orc_vector_average(f0[0].v, f0[0].n);
                    if (n2 > 0) a2 = orc_vector_average(f0[1].v, f0[1].n);
                    if (n3 > 0) a3 = orc_vector_average(f0[2].v, f0[2].n);
                    fprintf(outFW[i], "%s\t%d\t%d\t%g\t%g\t%g\t%d\t%d\t%d\n", chr, currentWindowStart, currentWindowEnd, a1, a2, a3, n1, n2, n3);
                    dv_clear(&f0[0]); dv_clear(&f0[1]); dv_clear(&f0[2]);
                }
                if (coordDouble > currentWindowEnd) { currentWindowStart += fixedWindowSize; currentWindowEnd += fixedWindowSize; }
                else if (coordDouble < currentWindowStart) { currentWindowStart = 0; currentWindowEnd = fixedWindowSize; }
            }
            for (int i = 0; i < n_trios; i++) {                                                 /* :307-343, ABS.cpp:243-277 */
                double pq[4], nq[4];
                int ok = 1, all0 = 1, all1 = 1;
                for (int x = 0; x < arity && ok; x++) {
                    pq[x] = aaf[trio[4 * i + x]]; nq[x] = alleles[trio[4 * i + x]];
                    if (pq[x] == -1) ok = 0;
                    if (pq[x] != 0) all0 = 0;
                    if (pq[x] != 1) all1 = 0;
                }
                if (!ok || all0 || all1) continue;
                usedVars[i]++;
                double val[3];
                if (arity == 3) orc_pbs_from_afs(pq[0], pq[1], pq[2], nq[0], nq[1], nq[2], val);
                else orc_abs_from_afs(pq[0], pq[1], pq[2], pq[3], nq[0], nq[1], nq[2], nq[3], val);
                double *d = dq + (size_t)i * 4 * W;
                int h = head[i];
                for (int x = 0; x < nval; x++) d[(size_t)x * W + (size_t)h] = val[x];
                d[3 * W + (size_t)h] = coordDouble;
                head[i] = h = (h + 1) % (int)W;                                                  /* push_back + pop_front */
                if (arity == 3) for (int x = 0; x < 3; x++) dv_push(&fw[3 * i + x], val[x]);
                if (an && snpGene[0]) {                                                           /* :333-341 */
                    int cat = strcmp(snpCategory, "exon") == 0 ? 0 : strcmp(snpCategory, "intron") == 0 ? 1 : strcmp(snpCategory, "promoter") == 0 ? 2 : -1;
                    if (cat >= 0) for (int x = 0; x < 3; x++) dv_push(&geneRes[9 * i + 3 * cat + x], val[x]);
                }
                if (usedVars[i] > windowSize && (usedVars[i] % windowStep == 0)) {
                    double av[3];
                    for (int x = 0; x < nval; x++) {
                        double sum = 0;
                        for (size_t t = 0; t < W; t++) sum += d[(size_t)x * W + (size_t)((h + (int)t) % (int)W)];
                        av[x] = sum / (double)W;
                    }
                    if (arity == 3) fprintf(outW[i], "%s\t%d\t%s\t%g\t%g\t%g\n", chr, (int)d[3 * W + (size_t)h], coord, av[0], av[1], av[2]);
                    else fprintf(outW[i], "%s\t%g\t%s\t%g\t%g\n", chr, d[3 * W + (size_t)h], coord, av[0], av[1]);   /* ABS.cpp:275: the coordinate stays a double */
                }
            }
            if (an && previousGene[0] && strcmp(currentGene, previousGene) != 0) {                /* :348-361 */
                for (int i = 0; i < n_trios; i++) {
                    dvec *g = &geneRes[9 * i];
                    int cnt[3] = {(int)g[0].n, (int)g[3].n, (int)g[6].n};
                    fprintf(outGenes[i], "%s\t%d\t%d\t%d", previousGene, cnt[0], cnt[1], cnt[2]);
                    for (int y = 0; y < 9; y++) fprintf(outGenes[i], "\t%g", cnt[y / 3] > 0 ? orc_vector_average(g[y].v, g[y].n) : 0.0);
                    fprintf(outGenes[i], "\n");
                    for (int y = 0; y < 9; y++) dv_clear(&g[y]);
                }
                free(previousGene); previousGene = strdup(currentGene);
            }
        }
    }
    if (an) {
        for (int i = 0; i < n_trios; i++) if (outGenes[i]) fclose(outGenes[i]);
        for (int i = 0; i < 9 * n_trios; i++) dv_free(&geneRes[i]);
        free(geneRes); orc_annot_free(an);
    }
    free(outGenes); free(currentGene); free(previousGene);
    for (int i = 0; i < n_trios; i++) { if (outW[i]) fclose(outW[i]); if (outFW[i]) fclose(outFW[i]); }
    for (int i = 0; i < 3 * n_trios; i++) dv_free(&fw[i]);
    lr_close(&vcf);
    free(fw); free(dq); free(head); free(usedVars); free(trio); free(outW); free(outFW);
    free(fields); free(sample_pop); free(alt); free(alleles); free(allelesC); free(aaf); free(aafC); free(use_pop);
    for (int k = 0; k < n_pops; k++) {
        for (int i = 0; i < pops[k].n_ids; i++) free(pops[k].ids[i]);
        free(pops[k].ids); free(pops[k].pos); free(pops[k].name);
    }
    free(pops);
    for (int i = 0; i < n_ids; i++) { free(id_names[i]); free(id_pop[i]); }
    free(id_names); free(id_pop);
    return rc;
}

int orc_pbs_run(const char *vcf_path, const char *pops_path, const char *trios_path,
                int windowSize, int windowStep, int fixedWindowSize, const char *run_name, const char *out_dir) {
    return branch_run(3, vcf_path, pops_path, trios_path, windowSize, windowStep, fixedWindowSize, run_name, out_dir);
}
/* with --annot (annot_path may be NULL or ""): P1_P2_P3_PBSGenes_<run>_<SIZE>_<STEP>.txt per trio */
int orc_pbs_run_annot(const char *vcf_path, const char *pops_path, const char *trios_path, int windowSize, int windowStep,
                      int fixedWindowSize, const char *run_name, const char *out_dir, const char *annot_path) {
    g_pbs_annot_path = annot_path && annot_path[0] ? annot_path : NULL;
    int rc = branch_run(3, vcf_path, pops_path, trios_path, windowSize, windowStep, fixedWindowSize, run_name, out_dir);
    g_pbs_annot_path = NULL;
    return rc;
}
/* -i / --allow-indels-and-multiallelics for the drivers that have it (PBS, DistOutgroups): only ALT == "*" is skipped;
 * every other line is counted by the characters '0' and '1' of its genotypes (getBasicCounts) */
void orc_set_allow_indels(int on) { g_allow_indels = on; }
/* --af of PBS (takes an argument that is not read, evo_PBS.cpp:48, 393) */
void orc_set_pbs_af(int on) { g_pbs_af = on; }

int orc_abs_run(const char *vcf_path, const char *pops_path, const char *quartets_path,
                int windowSize, int windowStep, const char *run_name, const char *out_dir) {
    return branch_run(4, vcf_path, pops_path, quartets_path, windowSize, windowStep, 10000, run_name, out_dir);
}

/* ===================================================================== */
/* `evo alleleFreq [-n] VCF POPS` (AFmain, evo_AlleleFeq.cpp:46-166)     */
/* without --use-genotype-probabilities.  The reference writes next to   */
/* the populations file; here the file goes to out_dir under the same    */
/* base name.                                                            */
/* ===================================================================== */
int orc_af_run(const char *vcf_path, const char *pops_path, const char *run_name, const char *out_dir) {
    lreader vcf;
    if (lr_open(&vcf, vcf_path) != 0) { fprintf(stderr, "Error: could not open %s\n", vcf_path); return -1; }
    FILE *sf = fopen(pops_path, "r");
    if (!sf) { lr_close(&vcf); return -2; }
    /* :62-68  popToIDsMap / IDsToPopMap (a later line with the same ID overwrites its pop) */
    pop_t *pops = NULL; int n_pops = 0, pcap = 0;
    char **id_names = NULL; char **id_pop = NULL; int n_ids = 0, icap = 0;
    {
        char *l = NULL; size_t cap = 0; ssize_t r;
        while ((r = getline(&l, &cap, sf)) >= 0) {
            if (r && l[r - 1] == '\n') l[r - 1] = '\0';
            char *tab = strchr(l, '\t');
            if (!tab) continue;
            *tab = '\0';
            char *id = l, *pn = tab + 1;
            char *tab2 = strchr(pn, '\t'); if (tab2) *tab2 = '\0';
            int k = -1;
            for (int i = 0; i < n_pops; i++) if (strcmp(pops[i].name, pn) == 0) { k = i; break; }
            if (k < 0) {
                if (n_pops == pcap) { pcap = pcap ? pcap * 2 : 16; pops = (pop_t *)realloc(pops, sizeof(pop_t) * (size_t)pcap); }
                k = n_pops++;
                memset(&pops[k], 0, sizeof(pop_t));
                pops[k].name = strdup(pn);
            }
            pop_t *P = &pops[k];
            if (P->n_ids == P->cap) { P->cap = P->cap ? P->cap * 2 : 16; P->ids = (char **)realloc(P->ids, sizeof(char *) * (size_t)P->cap); }
            P->ids[P->n_ids++] = strdup(id);
            int found = -1;
            for (int i = 0; i < n_ids; i++) if (strcmp(id_names[i], id) == 0) { found = i; break; }
            if (found < 0) {
                if (n_ids == icap) { icap = icap ? icap * 2 : 64; id_names = (char **)realloc(id_names, sizeof(char *) * (size_t)icap); id_pop = (char **)realloc(id_pop, sizeof(char *) * (size_t)icap); }
                id_names[n_ids] = strdup(id); id_pop[n_ids] = strdup(pn); n_ids++;
            } else { free(id_pop[found]); id_pop[found] = strdup(pn); }
        }
        free(l); fclose(sf);
    }
    qsort(pops, (size_t)n_pops, sizeof(pop_t), cmp_pop);    /* std::map order */
    fprintf(stderr, "There are %d populations \n", n_pops);
    /* :82 stripExtension(setsFile) + "_" + runName + "_AF.txt" */
    const char *slash = strrchr(pops_path, '/');
    char *base = strdup(slash ? slash + 1 : pops_path);
    { char *dot = strrchr(base, '.'); if (dot) *dot = '\0'; }
    char name[768];
    snprintf(name, sizeof name, "%s_%s_AF.txt", base, run_name);
    free(base);
    char *op = path_join(out_dir, name); FILE *out = fopen(op, "w"); free(op);
    if (!out) { lr_close(&vcf); return -3; }

    uint8_t *use_pop = (uint8_t *)calloc((size_t)n_pops + 1, 1);
    int32_t *sample_pop = NULL; int n_samples = 0;
    char **fields = NULL; int fcap = 0;
    int *alt = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1)), *alleles = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1));
    int *allelesC = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1));
    double *aaf = (double *)malloc(sizeof(double) * (size_t)(n_pops + 1)), *aafC = (double *)malloc(sizeof(double) * (size_t)(n_pops + 1));
    int totalVariantNumber = 0, rc = 0;
    char *line;
    while ((line = lr_getline(&vcf)) != NULL) {
        { char *w = line; for (char *r = line; *r; r++) if (*r != '\r') *w++ = *r; *w = '\0'; }   /* :85 */
        if (line[0] == '#' && line[1] == '#') continue;
        else if (line[0] == '#' && line[1] == 'C') {
            char *hdr = strdup(line); char **hf = NULL; int hcap = 0;
            int nh = split_inplace(hdr, '\t', &hf, &hcap);
            n_samples = nh - 9; if (n_samples < 0) n_samples = 0;
            sample_pop = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_samples + 1));
            for (int i = 0; i < n_samples; i++) {            /* :92-94 */
                sample_pop[i] = -1;
                for (int j = 0; j < n_ids; j++)
                    if (strcmp(id_names[j], hf[9 + i]) == 0) {
                        for (int k = 0; k < n_pops; k++) if (strcmp(pops[k].name, id_pop[j]) == 0) sample_pop[i] = k;
                        break;
                    }
            }
            for (int k = 0; k < n_pops; k++) {               /* :97-108 */
                pops[k].pos = (size_t *)malloc(sizeof(size_t) * (size_t)(pops[k].n_ids + 1));
                pops[k].n_pos = locate_set(hf + 9, n_samples, pops[k].ids, pops[k].n_ids, pops[k].pos);
                if (pops[k].n_pos == 0) {
                    fprintf(stderr, "Did not find any samples in the VCF for \"%s\"\n", pops[k].name);
                    rc = -4;
                }
            }
            free(hf); free(hdr);
            if (rc) break;
        } else {
            totalVariantNumber++;
            if (totalVariantNumber == 1) {                   /* :114-120 */
                fprintf(out, "chr\tcoord\tref\talt");
                for (int k = 0; k < n_pops; k++) fprintf(out, "\t%s", pops[k].name);
                fprintf(out, "\n");
            }
            int nf = split_inplace(line, '\t', &fields, &fcap);
            if (nf < 9) continue;
            if (orc_general_site_flags(fields[3], fields[4]) & ORC_SITE_SKIP) continue;   /* :124-127 */
            orc_general_counts((const char *const *)(fields + 9), n_samples, sample_pop, n_pops, use_pop,
                               alt, alleles, aaf, allelesC, aafC);
            fprintf(out, "%s\t%s\t%s\t%s", fields[0], fields[1], fields[3], fields[4]);
            for (int k = 0; k < n_pops; k++) fprintf(out, "\t%g", aaf[k]);   /* setAAFs in map order, :149-151 */
            fprintf(out, "\n");
        }
    }
    fclose(out); lr_close(&vcf);
    free(fields); free(sample_pop); free(alt); free(alleles); free(allelesC); free(aaf); free(aafC); free(use_pop);
    for (int k = 0; k < n_pops; k++) {
        for (int i = 0; i < pops[k].n_ids; i++) free(pops[k].ids[i]);
        free(pops[k].ids); free(pops[k].pos); free(pops[k].name);
    }
    free(pops);
    for (int i = 0; i < n_ids; i++) { free(id_names[i]); free(id_pop[i]); }
    free(id_names); free(id_pop);
    return rc;
}

/* ===================================================================== */
/* checker for the kernels' reciprocal-table division (no reference      */
/* counterpart): counts (c, nm) pairs for which                          */
/*   q0 = a*r; e = fma(-q0,b,a); q = fma(e,r,q0),  r = RN(1/b)           */
/* differs from IEEE a/b, for a/b = c/nm and (p(1-p))/(nm-1).            */
/* ===================================================================== */
static double rcp_div(double a, double b, double r) {
    double q0 = a * r;
    double e = fma(-q0, b, a);
    return fma(e, r, q0);
}
long orc_check_rcp_division(int max_n) {
    long bad = 0;
    for (int nm = 0; nm <= max_n; nm++) {
        double r = 1.0 / (double)nm, r1 = 1.0 / (double)(nm > 0 ? nm - 1 : 0);
        for (int c = 0; c <= nm; c++) {
            double p = (double)c / (double)nm, p2 = rcp_div((double)c, (double)nm, r);
            if (memcmp(&p, &p2, 8) != 0 && !(isnan(p) && isnan(p2))) bad++;
            double t = p * (1.0 - p);
            double h = t / (double)(nm - 1), h2 = rcp_div(t, (double)(nm - 1), r1);
            if (memcmp(&h, &h2, 8) != 0 && !(isnan(h) && isnan(h2))) bad++;
        }
    }
    return bad;
}

/* ===================================================================== */
/* DistOutgroups (evo_distanceToOutgroups.cpp)                           */
/* ===================================================================== */
/* the populations file and the header mapping every GeneralSetCounts driver starts with, for the drivers below */
typedef struct {
    pop_t *pops; int n_pops;                 /* sorted by name: std::map order */
    char **id_names, **id_pop; int n_ids;    /* IDsToPopMap (a later line with the same ID overwrites its population) */
} popfile_t;

static int popfile_read(const char *path, popfile_t *pf) {
    memset(pf, 0, sizeof *pf);
    FILE *sf = fopen(path, "r");
    if (!sf) return -2;
    int pcap = 0, icap = 0;
    char *l = NULL; size_t cap = 0; ssize_t r;
    while ((r = getline(&l, &cap, sf)) >= 0) {
        if (r && l[r - 1] == '\n') l[r - 1] = '\0';
        char *tab = strchr(l, '\t');
        if (!tab) continue;
        *tab = '\0';
        char *id = l, *pn = tab + 1;
        char *tab2 = strchr(pn, '\t'); if (tab2) *tab2 = '\0';
        int k = -1;
        for (int i = 0; i < pf->n_pops; i++) if (strcmp(pf->pops[i].name, pn) == 0) { k = i; break; }
        if (k < 0) {
            if (pf->n_pops == pcap) { pcap = pcap ? pcap * 2 : 16; pf->pops = (pop_t *)realloc(pf->pops, sizeof(pop_t) * (size_t)pcap); }
            k = pf->n_pops++;
            memset(&pf->pops[k], 0, sizeof(pop_t));
            pf->pops[k].name = strdup(pn);
        }
        pop_t *P = &pf->pops[k];
        if (P->n_ids == P->cap) { P->cap = P->cap ? P->cap * 2 : 16; P->ids = (char **)realloc(P->ids, sizeof(char *) * (size_t)P->cap); }
        P->ids[P->n_ids++] = strdup(id);
        int found = -1;
        for (int i = 0; i < pf->n_ids; i++) if (strcmp(pf->id_names[i], id) == 0) { found = i; break; }
        if (found < 0) {
            if (pf->n_ids == icap) {
                icap = icap ? icap * 2 : 64;
                pf->id_names = (char **)realloc(pf->id_names, sizeof(char *) * (size_t)icap);
                pf->id_pop = (char **)realloc(pf->id_pop, sizeof(char *) * (size_t)icap);
            }
            pf->id_names[pf->n_ids] = strdup(id); pf->id_pop[pf->n_ids] = strdup(pn); pf->n_ids++;
        } else { free(pf->id_pop[found]); pf->id_pop[found] = strdup(pn); }
    }
    free(l); fclose(sf);
    qsort(pf->pops, (size_t)pf->n_pops, sizeof(pop_t), cmp_pop);
    return 0;
}

/* #CHROM line -> population of every genotype column (posToPopMap) and the columns of every population
 * (popToPosMap); returns the number of genotype columns, *rc = -4 if a population has no column */
static int popfile_map_header(popfile_t *pf, const char *line, int32_t **sample_pop, int *rc) {
    char *hdr = strdup(line); char **hf = NULL; int hcap = 0;
    int nh = split_inplace(hdr, '\t', &hf, &hcap);
    int n_samples = nh - 9; if (n_samples < 0) n_samples = 0;
    *sample_pop = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_samples + 1));
    for (int i = 0; i < n_samples; i++) {
        (*sample_pop)[i] = -1;
        for (int j = 0; j < pf->n_ids; j++)
            if (strcmp(pf->id_names[j], hf[9 + i]) == 0) {
                for (int k = 0; k < pf->n_pops; k++) if (strcmp(pf->pops[k].name, pf->id_pop[j]) == 0) (*sample_pop)[i] = k;
                break;
            }
    }
    for (int k = 0; k < pf->n_pops; k++) {
        pf->pops[k].pos = (size_t *)malloc(sizeof(size_t) * (size_t)(pf->pops[k].n_ids + 1));
        pf->pops[k].n_pos = locate_set(hf + 9, n_samples, pf->pops[k].ids, pf->pops[k].n_ids, pf->pops[k].pos);
        if (pf->pops[k].n_pos == 0) { fprintf(stderr, "Did not find any samples in the VCF for \"%s\"\n", pf->pops[k].name); *rc = -4; }
    }
    free(hf); free(hdr);
    return n_samples;
}

static void popfile_free(popfile_t *pf) {
    for (int k = 0; k < pf->n_pops; k++) {
        for (int i = 0; i < pf->pops[k].n_ids; i++) free(pf->pops[k].ids[i]);
        free(pf->pops[k].ids); free(pf->pops[k].pos); free(pf->pops[k].name);
    }
    free(pf->pops);
    for (int i = 0; i < pf->n_ids; i++) { free(pf->id_names[i]); free(pf->id_pop[i]); }
    free(pf->id_names); free(pf->id_pop);
}

/* lines of a file as they are (getline without any trimming, evo_distanceToOutgroups.cpp:104-117) */
static char **read_lines(const char *path, int *n) {
    *n = 0;
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    char **out = NULL; int capn = 0;
    char *l = NULL; size_t cap = 0; ssize_t r;
    while ((r = getline(&l, &cap, f)) >= 0) {
        if (r && l[r - 1] == '\n') l[r - 1] = '\0';
        if (*n == capn) { capn = capn ? capn * 2 : 16; out = (char **)realloc(out, sizeof(char *) * (size_t)capn); }
        out[(*n)++] = strdup(l);
    }
    free(l); fclose(f);
    if (!out) out = (char **)calloc(1, sizeof(char *));
    return out;
}

/* DxyPerSNPfromAFs, evo_distanceToOutgroups.cpp:62-65 */
double orc_dxy_from_afs(double AF1, double AF2) { return AF1 * (1 - AF2) + AF2 * (1 - AF1); }

/* The accumulation of DistOutMain (evo_distanceToOutgroups.cpp:274-294) between two window emissions, on counts:
 * window w covers sites [win_off[w], win_off[w+1]); per window and outgroup i: used / missing site counts
 * ([n_windows][n_out]); per window, outgroup and ingroup j ([n_windows][n_out][n_in]): the sum of the per-site dxy in
 * site order, the number of sites in it, and the number of sites where only the ingroup is missing.
 * carry_first != 0: the entries of window 0 hold the running values of a window begun earlier and are continued. */
void orc_dist_windows(const int32_t *alt, const int32_t *nm, int n_pops, const uint8_t *site_flags,
                      const int64_t *win_off, int n_windows, const int32_t *out_idx, int n_out, const int32_t *in_idx, int n_in,
                      int carry_first, double *sum, int32_t *nsnp, int32_t *miss, int32_t *used, int32_t *missing_vars) {
    for (int w = 0; w < n_windows; w++)
        for (int i = 0; i < n_out; i++) {
            size_t wi = (size_t)w * (size_t)n_out + (size_t)i;
            if (!(carry_first && w == 0)) {
                used[wi] = 0; missing_vars[wi] = 0;
                for (int j = 0; j < n_in; j++) { sum[wi * (size_t)n_in + (size_t)j] = 0; nsnp[wi * (size_t)n_in + (size_t)j] = 0; miss[wi * (size_t)n_in + (size_t)j] = 0; }
            }
            for (int64_t s = win_off[w]; s < win_off[w + 1]; s++) {
                if (site_flags && (site_flags[s] & ORC_SITE_SKIP)) continue;
                int mo = nm[s * n_pops + out_idx[i]];
                if (mo == 0) { missing_vars[wi]++; continue; }                 /* AFout == -1, :278-282 */
                used[wi]++;
                double AFout = (double)alt[s * n_pops + out_idx[i]] / mo;
                for (int j = 0; j < n_in; j++) {
                    int mi = nm[s * n_pops + in_idx[j]];
                    size_t o = wi * (size_t)n_in + (size_t)j;
                    if (mi != 0) { sum[o] += orc_dxy_from_afs(AFout, (double)alt[s * n_pops + in_idx[j]] / mi); nsnp[o]++; }
                    else miss[o]++;
                }
            }
        }
}

/* `evo DistOutgroups [-f FW] [-n RUN] [--accessibleGenomeBED=BED] VCF POPS OUTGROUPS INGROUPS` (DistOutMain,
 * evo_distanceToOutgroups.cpp:68-306) without -i.  Output files go to out_dir; bed_path may be NULL. */
int orc_distout_run(const char *vcf_path, const char *pops_path, const char *outgroups_path, const char *ingroups_path,
                    int fixedWindowSize, const char *run_name, const char *out_dir) {
    return orc_distout_run_bed(vcf_path, pops_path, outgroups_path, ingroups_path, fixedWindowSize, run_name, out_dir, NULL);
}

int orc_distout_run_bed(const char *vcf_path, const char *pops_path, const char *outgroups_path, const char *ingroups_path,
                        int fixedWindowSize, const char *run_name, const char *out_dir, const char *bed_path) {
    struct orc_bed *ag = NULL;                                               /* :121-127 */
    if (bed_path && *bed_path) { ag = orc_bed_load(bed_path); if (!ag) { fprintf(stderr, "Error: could not open %s\n", bed_path); return -1; } }
    lreader vcf;
    if (lr_open(&vcf, vcf_path) != 0) { fprintf(stderr, "Error: could not open %s\n", vcf_path); return -1; }
    popfile_t pf;
    int rc = popfile_read(pops_path, &pf);
    if (rc) { lr_close(&vcf); return rc; }
    fprintf(stderr, "There are %d populations \n", pf.n_pops);
    int n_in = 0, n_out = 0;
    char **ingroups = read_lines(ingroups_path, &n_in);                      /* :104-106 */
    char **outgroups = read_lines(outgroups_path, &n_out);                   /* :110-117 */
    if (!ingroups || !outgroups) { lr_close(&vcf); popfile_free(&pf); return -2; }
    FILE **outFW = (FILE **)calloc((size_t)n_out + 1, sizeof(FILE *));
    int *out_k = (int *)malloc(sizeof(int) * (size_t)(n_out + 1)), *in_k = (int *)malloc(sizeof(int) * (size_t)(n_in + 1));
    for (int i = 0; i < n_out; i++) {
        char name[1024];
        snprintf(name, sizeof name, "%s_DIST_%s_FW%d.txt", outgroups[i], run_name, fixedWindowSize);
        char *pth = path_join(out_dir, name); outFW[i] = fopen(pth, "w"); free(pth);
        if (!outFW[i]) { rc = -3; break; }
        fprintf(outFW[i], "chr\twStart\twEnd\tSNPs_used\tSNPs_missing\tAccessibleSizeBP\t");
        for (int j = 0; j < n_in; j++) fprintf(outFW[i], "%s%s", ingroups[j], j == n_in - 1 ? "\n" : "\t");   /* print_vector */
        out_k[i] = -1;
        for (int k = 0; k < pf.n_pops; k++) if (strcmp(pf.pops[k].name, outgroups[i]) == 0) out_k[i] = k;
        if (out_k[i] < 0) rc = -6;                                           /* setAAFs.at() would throw */
    }
    for (int j = 0; j < n_in; j++) {
        in_k[j] = -1;
        for (int k = 0; k < pf.n_pops; k++) if (strcmp(pf.pops[k].name, ingroups[j]) == 0) in_k[j] = k;
        if (in_k[j] < 0 && n_out > 0) rc = -6;
    }
    /* :133-149 */
    dvec *perSNP = (dvec *)calloc((size_t)(n_out * n_in) + 1, sizeof(dvec));
    int *missingDist = (int *)calloc((size_t)(n_out * n_in) + 1, sizeof(int));
    double *averaged = (double *)calloc((size_t)(n_out * n_in) + 1, sizeof(double));
    int *usedVars = (int *)calloc((size_t)n_out + 1, sizeof(int)), *missingVars = (int *)calloc((size_t)n_out + 1, sizeof(int));
    int currentWindowStart = 0, currentWindowEnd = currentWindowStart + fixedWindowSize;
    int n_pops = pf.n_pops;
    uint8_t *use_pop = (uint8_t *)calloc((size_t)n_pops + 1, 1);
    int32_t *sample_pop = NULL; int n_samples = 0;
    char **fields = NULL; int fcap = 0;
    int *alt = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1)), *alleles = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1));
    int *allelesC = (int *)malloc(sizeof(int) * (size_t)(n_pops + 1));
    double *aaf = (double *)malloc(sizeof(double) * (size_t)(n_pops + 1)), *aafC = (double *)malloc(sizeof(double) * (size_t)(n_pops + 1));
    char *line;
    while (!rc && (line = lr_getline(&vcf)) != NULL) {
        { char *w = line; for (char *r = line; *r; r++) if (*r != '\r') *w++ = *r; *w = '\0'; }   /* :156 */
        if (line[0] == '#' && line[1] == '#') continue;
        else if (line[0] == '#' && line[1] == 'C') {
            n_samples = popfile_map_header(&pf, line, &sample_pop, &rc);
        } else {
            int nf = split_inplace(line, '\t', &fields, &fcap);
            if (nf < 9) continue;
            const char *chr = fields[0];
            double coordDouble = strtod(fields[1], NULL);
            if (strcmp(fields[4], "*") == 0 || (!g_allow_indels && (strlen(fields[3]) > 1 || strlen(fields[4]) > 1))) continue;   /* :194-202 */
            orc_general_counts((const char *const *)(fields + 9), n_samples, sample_pop, n_pops, use_pop,
                               alt, alleles, aaf, allelesC, aafC);
            if (coordDouble > currentWindowEnd || coordDouble < currentWindowStart) {                    /* :221-269 */
                int accessibleInThisWindow = fixedWindowSize;
                if (ag) {                                                    /* :224-230: the scaffold of THIS line is asked */
                    if (coordDouble > currentWindowEnd) accessibleInThisWindow = orc_bed_accessible_bp(ag, chr, currentWindowStart, currentWindowStart + fixedWindowSize);
                    else if (coordDouble < currentWindowStart) accessibleInThisWindow = orc_bed_accessible_bp(ag, chr, 0, fixedWindowSize);
                }
                for (int i = 0; i < n_out; i++) {
                    for (int j = 0; j < n_in; j++) {
                        dvec *v = &perSNP[i * n_in + j];
                        int nSNPs = (int)v->n, md = missingDist[i * n_in + j];
                        if (nSNPs > 0) {
                            double missingProportion = (double)md / (nSNPs + md);
                            if (missingProportion > 0.5) averaged[i * n_in + j] = NAN;
                            else {
                                double a = orc_vector_average_with_region(v->v, v->n, accessibleInThisWindow);
                                double nonMissingProportion = (double)nSNPs / (nSNPs + md);
                                double scalingFactor = 1 + (missingProportion / nonMissingProportion);
                                averaged[i * n_in + j] = a * scalingFactor;
                            }
                        } else averaged[i * n_in + j] = NAN;
                        dv_clear(v);
                    }
                    fprintf(outFW[i], "%s\t%d\t%d\t%d\t%d\t%d\t", chr, currentWindowStart, currentWindowEnd, usedVars[i], missingVars[i], accessibleInThisWindow);
                    for (int j = 0; j < n_in; j++) fprintf(outFW[i], "%g%s", averaged[i * n_in + j], j == n_in - 1 ? "\n" : "\t");
                }
                for (int i = 0; i < n_out; i++) {
                    usedVars[i] = 0; missingVars[i] = 0;
                    for (int j = 0; j < n_in; j++) missingDist[i * n_in + j] = 0;
                }
                if (coordDouble > currentWindowEnd) { currentWindowStart += fixedWindowSize; currentWindowEnd += fixedWindowSize; }
                else if (coordDouble < currentWindowStart) { currentWindowStart = 0; currentWindowEnd = fixedWindowSize; }
            }
            for (int i = 0; i < n_out; i++) {                                                            /* :274-294 */
                double AFout = aaf[out_k[i]];
                if (AFout == -1) { missingVars[i]++; continue; }
                usedVars[i]++;
                for (int j = 0; j < n_in; j++) {
                    double AFin = aaf[in_k[j]];
                    if (AFin != -1) dv_push(&perSNP[i * n_in + j], orc_dxy_from_afs(AFout, AFin));
                    else missingDist[i * n_in + j]++;
                }
            }
        }
    }
    for (int i = 0; i < n_out; i++) if (outFW[i]) fclose(outFW[i]);
    for (int i = 0; i < n_out * n_in; i++) dv_free(&perSNP[i]);
    lr_close(&vcf);
    free(perSNP); free(missingDist); free(averaged); free(usedVars); free(missingVars); free(outFW); free(out_k); free(in_k);
    free(fields); free(sample_pop); free(alt); free(alleles); free(allelesC); free(aaf); free(aafC); free(use_pop);
    for (int i = 0; i < n_in; i++) free(ingroups[i]);
    for (int i = 0; i < n_out; i++) free(outgroups[i]);
    free(ingroups); free(outgroups);
    popfile_free(&pf);
    if (ag) orc_bed_free(ag);
    return rc;
}

/* ===================================================================== */
/* stats --diff-matrix (process_vcf_stats.cpp, process_vcf_stats_functions.cpp) */
/* ===================================================================== */
/* getThisVariantCountsSimple, process_vcf_utils.cpp:358-408, for the fields the difference matrices read.
 * genotypes = the n genotype columns.  randomisePhase (:387) only swaps the two alleles of an unphased individual;
 * every consumer restated here is symmetric in them, so it is left out.  haps[2n], ind_with_variant[n], missing[n]. */
void orc_stats_variant_counts(const char *alt_field, const char *const *genotypes, int n,
                              int *ind_with_variant, int *haps, uint8_t *missing, int *is_multiallelic) {
    /* split(fields[4], ','): a trailing empty field is dropped, an empty string gives no field */
    int numAltAlleles = 0; const char *last = alt_field; size_t last_len = 0;
    {
        const char *a = alt_field;
        if (*a) for (;;) {
            const char *e = strchr(a, ',');
            size_t l = e ? (size_t)(e - a) : strlen(a);
            numAltAlleles++; last = a; last_len = l;
            if (!e) break;
            a = e + 1;
            if (!*a) break;
        }
    }
    int alleleAsMissing = -1, multi = 0;
    if (numAltAlleles > 1) {                                           /* :369-381 */
        if (last_len == 1 && last[0] == '*') { alleleAsMissing = numAltAlleles; if ((numAltAlleles - 1) > 1) multi = 1; }
        else multi = 1;
    }
    for (int i = 0; i < n; i++) {
        char v1 = genotypes[i][0], v2 = genotypes[i][2];
        int v1int = v1 - '0', v2int = v2 - '0';
        ind_with_variant[i] = 0; haps[2 * i] = 0; haps[2 * i + 1] = 0; missing[i] = 0;
        if (v1 == '.' || v2 == '.' || v1int == alleleAsMissing || v2int == alleleAsMissing) missing[i] = 1;
        else { haps[2 * i] = v1int; haps[2 * i + 1] = v2int; }
        if (!multi) {                                                  /* :397-406 */
            if (v1 == '1') ind_with_variant[i]++;
            if (v2 == '1') ind_with_variant[i]++;
        }
    }
    *is_multiallelic = multi;
}

static const double ORC_DIFF_BOTH_HETS_ME = 0.5, ORC_DIFF_BOTH_HETS_RICHARD = (2.0 / 3.0), ORC_DIFF_HOM_ME = 1,
                    ORC_DIFF_HOM_RICHARD = (2.0 / 3.0), ORC_DIFF_HOM_HET = 0.5;   /* stats_functions.cpp:12-16 */

/* diffs_between_individuals (stats_functions.cpp:180-232) and diffs_between_individuals_with_multialleleics
 * (:235-281), literally, over S sites: all matrices are n x n row-major and are ADDED to (bootstrap copies may be
 * NULL).  ind_with_variant [S][n], haps [S][2n], missing [S][n], multi [S]. */
void orc_indiv_diffs(int n, int64_t S, const int *ind_with_variant, const int *haps, const uint8_t *missing, const uint8_t *multi,
                     double *diffs, double *diffs_me, double *diffs_me_bootstrap, double *hets_vs_homs,
                     int *pairwise_missingness, int *pairwise_missingness_bootstrap) {
    for (int64_t s = 0; s < S; s++) {
        const int *iv = ind_with_variant + s * n, *hp = haps + s * 2 * n;
        const uint8_t *ms = missing + s * n;
        for (int i = 0; i < n; i++) {
            for (int j = 0; j <= i; j++) {
                size_t o = (size_t)i * (size_t)n + (size_t)j;
                if (ms[i] || ms[j]) {
                    pairwise_missingness[o]++;
                    if (pairwise_missingness_bootstrap) pairwise_missingness_bootstrap[o]++;
                    continue;
                }
                if (!multi[s]) {
                    int ind_i = iv[i], ind_j = iv[j];
                    if (j < i) {
                        double r, me;
                        if (ind_i == 1 && ind_j == 1) { r = ORC_DIFF_BOTH_HETS_RICHARD; me = ORC_DIFF_BOTH_HETS_ME; hets_vs_homs[o]++; }
                        else if ((ind_i == 2 && ind_j == 0) || (ind_i == 0 && ind_j == 2)) {
                            r = ORC_DIFF_HOM_RICHARD; me = ORC_DIFF_HOM_ME; hets_vs_homs[(size_t)j * (size_t)n + (size_t)i]++;
                        } else if ((ind_i == 2 && ind_j == 1) || (ind_i == 1 && ind_j == 2) || (ind_i == 0 && ind_j == 1) || (ind_i == 1 && ind_j == 0)) {
                            r = ORC_DIFF_HOM_HET; me = ORC_DIFF_HOM_HET;
                        } else { r = 0; me = 0; }
                        diffs[o] = diffs[o] + r;
                        diffs_me[o] = diffs_me[o] + me;
                        if (diffs_me_bootstrap) diffs_me_bootstrap[o] = diffs_me_bootstrap[o] + me;
                    } else if (ind_i == 1) {
                        diffs[o]++; diffs_me[o]++;
                        if (diffs_me_bootstrap) diffs_me_bootstrap[o]++;
                    }
                } else {
                    int ind_i = hp[2 * i], ind_i2 = hp[2 * i + 1], ind_j = hp[2 * j], ind_j2 = hp[2 * j + 1];
                    if (j < i) {
                        double totalD = 0; const double numComparisons = 4;
                        if (ind_i != ind_j) totalD++;
                        if (ind_i != ind_j2) totalD++;
                        if (ind_i2 != ind_j) totalD++;
                        if (ind_i2 != ind_j2) totalD++;
                        double me = totalD / numComparisons;
                        diffs_me[o] = diffs_me[o] + me;
                        if (diffs_me_bootstrap) diffs_me_bootstrap[o] = diffs_me_bootstrap[o] + me;
                    } else if (ind_i != ind_i2) {
                        diffs_me[o]++;
                        if (diffs_me_bootstrap) diffs_me_bootstrap[o]++;
                    }
                }
            }
        }
    }
}

/* The same sites as five counts per pair j <= i (triangular index i(i+1)/2 + j), which is what the device returns:
 * hh both heterozygous (diagonal: heterozygous), hd opposite homozygotes, oh one homozygous one heterozygous -- all
 * three on bi-allelic sites; ms either individual missing (any site); d4 differing allele comparisons, of 4 per
 * multi-allelic site (diagonal: sites where the individual's own two alleles differ).  Counts are ADDED to. */
void orc_indiv_diff_counts(int n, int64_t S, const int *ind_with_variant, const int *haps, const uint8_t *missing, const uint8_t *multi,
                           int32_t *hh, int32_t *hd, int32_t *oh, int32_t *ms_out, int32_t *d4) {
    for (int64_t s = 0; s < S; s++) {
        const int *iv = ind_with_variant + s * n, *hp = haps + s * 2 * n;
        const uint8_t *ms = missing + s * n;
        for (int i = 0; i < n; i++)
            for (int j = 0; j <= i; j++) {
                size_t o = (size_t)i * ((size_t)i + 1) / 2 + (size_t)j;
                if (ms[i] || ms[j]) { ms_out[o]++; continue; }
                if (!multi[s]) {
                    int a = iv[i], b = iv[j];
                    if (j == i) { if (a == 1) hh[o]++; }
                    else if (a == 1 && b == 1) hh[o]++;
                    else if ((a == 2 && b == 0) || (a == 0 && b == 2)) hd[o]++;
                    else if ((a == 1) != (b == 1)) oh[o]++;
                } else if (j == i) { if (hp[2 * i] != hp[2 * i + 1]) d4[o]++; }
                else d4[o] += (hp[2 * i] != hp[2 * j]) + (hp[2 * i] != hp[2 * j + 1]) + (hp[2 * i + 1] != hp[2 * j]) + (hp[2 * i + 1] != hp[2 * j + 1]);
            }
    }
}

/* het_analysis (stats_functions.cpp:19-28) as statsMain calls it (process_vcf_stats.cpp:239-242: only at sites where
 * no genotype is missing), over S sites: counts.overall is the number of '1' characters of the line's genotypes
 * (0 on multi-allelic lines, utils.cpp:397-406), i.e. the sum of ind_with_variant.  Counts are ADDED to. */
void orc_indiv_hets(int n, int64_t S, const int *ind_with_variant, const uint8_t *missing, int32_t *het, int32_t *shared) {
    for (int64_t s = 0; s < S; s++) {
        const int *iv = ind_with_variant + s * n;
        const uint8_t *ms = missing + s * n;
        int anyMissing = 0, overall = 0;
        for (int i = 0; i < n; i++) { anyMissing |= ms[i] != 0; overall += iv[i]; }
        if (anyMissing) continue;
        for (int i = 0; i < n; i++)
            if (iv[i] == 1) {
                het[i]++;
                if (overall > 1) shared[i]++;
            }
    }
}

static void print_matrix_d(FILE *f, const double *m, int n, int precision) {      /* print_matrix, utils.h:326-335 */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) fprintf(f, "%.*g%s", precision, m[(size_t)i * (size_t)n + (size_t)j], j == n - 1 ? "\n" : "\t");
}

/* `evo stats --diff-matrix [--block-bootstrap=B] [--numAccessibleBP=A] VCF` (statsMain, process_vcf_stats.cpp:98-355,
 * the bDiffs and bDoBootstrap parts; B = 0: no bootstrap, A < 0: no scaling).  The four matrix files go to out_dir
 * under the VCF's base name, the bootstrap replicates too (the reference writes those next to the VCF). */
static int stats_run(const char *vcf_path, int doDiffs, int doHets, int bootstrapBlockSize, int numAccessibleBP, const char *out_dir);
int orc_stats_diffs_run(const char *vcf_path, int bootstrapBlockSize, int numAccessibleBP, const char *out_dir) {
    return stats_run(vcf_path, 1, 0, bootstrapBlockSize, numAccessibleBP, out_dir);
}
/* the same with --diff-matrix and / or --hets-per-individual (process_vcf_stats.cpp:240-242, 336-337; print_het_counts,
 * process_vcf_print_routines.cpp:29-56: <VCF base>.hets.txt and <VCF base>.sharedHets.txt, here in out_dir) */
int orc_stats_run(const char *vcf_path, int doDiffs, int doHets, int bootstrapBlockSize, int numAccessibleBP, const char *out_dir) {
    return stats_run(vcf_path, doDiffs, doHets, bootstrapBlockSize, numAccessibleBP, out_dir);
}
static int stats_run(const char *vcf_path, int doDiffs, int doHets, int bootstrapBlockSize, int numAccessibleBP, const char *out_dir) {
    lreader vcf;
    if (lr_open(&vcf, vcf_path) != 0) { fprintf(stderr, "Error: could not open %s\n", vcf_path); return -1; }
    /* fileRoot = stripExtension(fileName), fileNoPath = stripPath(fileRoot) (:103, :328) */
    char *root = strdup(vcf_path);
    { char *dot = strrchr(root, '.'); if (dot) *dot = '\0'; }
    const char *slash = strrchr(root, '/');
    const char *noPath = slash ? slash + 1 : root;
    int n = 0, totalVariantNumber = 0, bootstrapBlockNum = 0, rc = 0;
    long long phaseDraws = 0;
    char **names = NULL;
    double *diffs = NULL, *diffs_me = NULL, *hvh = NULL, *boot = NULL, **bootBlocks = NULL;
    int *miss = NULL, *bootMiss = NULL, **bootMissBlocks = NULL;
    int *iv = NULL, *hp = NULL; uint8_t *ms = NULL;
    int32_t *hetCounts = NULL, *hetsShared = NULL;
    char **fields = NULL; int fcap = 0;
    char *line;
    while ((line = lr_getline(&vcf)) != NULL) {
        if (line[0] == '#' && line[1] == '#') continue;
        else if (line[0] == '#' && line[1] == 'C') {
            int nf = split_inplace(line, '\t', &fields, &fcap);
            n = nf - 9; if (n < 0) n = 0;
            names = (char **)malloc(sizeof(char *) * (size_t)(n + 1));
            for (int i = 0; i < n; i++) names[i] = strdup(fields[9 + i]);
            size_t nn = (size_t)n * (size_t)n + 1;
            diffs = (double *)calloc(nn, sizeof(double)); diffs_me = (double *)calloc(nn, sizeof(double));
            hvh = (double *)calloc(nn, sizeof(double)); boot = (double *)calloc(nn, sizeof(double));
            miss = (int *)calloc(nn, sizeof(int)); bootMiss = (int *)calloc(nn, sizeof(int));
            iv = (int *)malloc(sizeof(int) * (size_t)(n + 1)); hp = (int *)malloc(sizeof(int) * (size_t)(2 * n + 2));
            ms = (uint8_t *)malloc((size_t)n + 1);
            hetCounts = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t)); hetsShared = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
        } else {
            if (!names) { rc = -5; break; }
            totalVariantNumber++;
            int nf = split_inplace(line, '\t', &fields, &fcap);
            if (nf < 9 + n) { rc = -5; break; }
            int multi = 0;
            orc_stats_variant_counts(fields[4], (const char *const *)(fields + 9), n, iv, hp, ms, &multi);
            if (n > 0 && fields[9][1] != '|')                       /* utils.cpp:384-387: randomisePhase draws from rand() */
                for (int i = 0; i < n; i++) if (fields[9 + i][0] != fields[9 + i][2]) phaseDraws++;
            uint8_t m8 = (uint8_t)multi;
            if (doHets) orc_indiv_hets(n, 1, iv, ms, hetCounts, hetsShared);                             /* :239-242 */
            if (doDiffs) orc_indiv_diffs(n, 1, iv, hp, ms, &m8, diffs, diffs_me, boot, hvh, miss, bootMiss);   /* :257-263 */
            if (bootstrapBlockSize > 0 && totalVariantNumber % bootstrapBlockSize == 0) {              /* :265-275 */
                size_t nn = (size_t)n * (size_t)n;
                bootBlocks = (double **)realloc(bootBlocks, sizeof(double *) * (size_t)(bootstrapBlockNum + 1));
                bootMissBlocks = (int **)realloc(bootMissBlocks, sizeof(int *) * (size_t)(bootstrapBlockNum + 1));
                bootBlocks[bootstrapBlockNum] = (double *)malloc(sizeof(double) * (nn + 1));
                bootMissBlocks[bootstrapBlockNum] = (int *)malloc(sizeof(int) * (nn + 1));
                memcpy(bootBlocks[bootstrapBlockNum], boot, sizeof(double) * nn);
                memcpy(bootMissBlocks[bootstrapBlockNum], bootMiss, sizeof(int) * nn);
                memset(boot, 0, sizeof(double) * nn); memset(bootMiss, 0, sizeof(int) * nn);
                bootstrapBlockNum++;
            }
        }
    }
    if (!rc && !names) rc = -5;
    if (!rc) {
        size_t nn = (size_t)n * (size_t)n;
        if (numAccessibleBP > -1) for (size_t k = 0; k < nn; k++) diffs_me[k] = (double)diffs_me[k] / numAccessibleBP;   /* :286-293 */
        if (bootstrapBlockSize > 0) {                                                                   /* :296-325 */
            /* the reference never seeds (a fresh process starts from 1) and randomisePhase (utils.cpp:268-276) has
             * drawn once per unphased genotype with two different allele characters by now */
            srand(1);
            for (long long k = 0; k < phaseDraws; k++) (void)rand();
            const int n_bootstrap_replicates = 100;
            double *rep = (double *)malloc(sizeof(double) * (nn + 1)), *repMiss = (double *)malloc(sizeof(double) * (nn + 1));
            for (int r = 0; r < n_bootstrap_replicates; r++) {
                char name[1024];
                snprintf(name, sizeof name, "%sboot.%d.txt", noPath, r);
                char *pth = path_join(out_dir, name); FILE *f = fopen(pth, "w"); free(pth);
                if (!f) { rc = -3; break; }
                for (size_t k = 0; k < nn; k++) { rep[k] = 0; repMiss[k] = 0; }
                for (int b = 0; b < bootstrapBlockNum; b++) {
                    int block = rand() % bootstrapBlockNum;
                    for (size_t k = 0; k < nn; k++) { rep[k] = rep[k] + bootBlocks[block][k]; repMiss[k] = repMiss[k] + bootMissBlocks[block][k]; }
                }
                int totalSitesBootstrap = bootstrapBlockNum * bootstrapBlockSize;
                for (size_t k = 0; k < nn; k++) {
                    double proportionUsed = 1 - (repMiss[k] / totalSitesBootstrap);
                    rep[k] = rep[k] / proportionUsed;
                }
                for (int i = 0; i < n; i++) fprintf(f, "%s%s", names[i], i == n - 1 ? "\n" : "\t");
                print_matrix_d(f, rep, n, 6);
                fclose(f);
            }
            free(rep); free(repMiss);
        }
        if (doHets) {                                        /* print_het_counts, print_routines.cpp:29-56; fileRoot keeps the path */
            char name[1024];
            snprintf(name, sizeof name, "%s.hets.txt", noPath);
            char *pth = path_join(out_dir, name); FILE *fh = fopen(pth, "w"); free(pth);
            snprintf(name, sizeof name, "%s.sharedHets.txt", noPath);
            pth = path_join(out_dir, name); FILE *fs = fopen(pth, "w"); free(pth);
            if (!fh || !fs) rc = -3;
            else {
                fprintf(fh, "# Het counts\n# Input file:%s.vcf\n", root);
                fprintf(fs, "# Shared het counts (line1) and proportions (line 2)\n# Input file:%s.vcf\n", root);
                for (int i = 0; i < n; i++) fprintf(fh, "%s%s", names[i], i == n - 1 ? "\n" : "\t");
                for (int i = 0; i < n; i++) fprintf(fh, "%d%s", hetCounts[i], i == n - 1 ? "\n" : "\t");
                for (int i = 0; i < n; i++) fprintf(fs, "%s%s", names[i], i == n - 1 ? "\n" : "\t");
                for (int i = 0; i < n; i++) fprintf(fs, "%d%s", hetsShared[i], i == n - 1 ? "\n" : "\t");
                for (int i = 0; i < n; i++) {                /* :45: 0 / 0 is the x86 default NaN, printed "-nan" */
                    volatile double num = (double)hetsShared[i], den = (double)hetCounts[i];
                    fprintf(fs, "%.6g%s", num / den, i == n - 1 ? "\n" : "\t");
                }
            }
            if (fh) fclose(fh);
            if (fs) fclose(fs);
        }
        if (doDiffs)
        for (int i = 0; i < n; i++)                          /* finalize_diffs_Hets_vs_Homs_proportions, utils.cpp:1148-1155 */
            for (int j = 0; j < i; j++) {
                hvh[(size_t)i * (size_t)n + (size_t)j] = hvh[(size_t)i * (size_t)n + (size_t)j] / hvh[(size_t)j * (size_t)n + (size_t)i];
                hvh[(size_t)j * (size_t)n + (size_t)i] = 0;
            }
        /* print_pairwise_diff_stats, print_routines.cpp:74-120 */
        const char *suffix[4] = {".diff_matrix.txt", ".diff_me_matrix.txt", ".hets_over_homs_matrix.txt", ".parwiseMissingness.txt"};
        FILE *f[4] = {NULL, NULL, NULL, NULL};
        for (int k = 0; doDiffs && k < 4; k++) {
            char name[1024];
            snprintf(name, sizeof name, "%s%s", noPath, suffix[k]);
            char *pth = path_join(out_dir, name); f[k] = fopen(pth, "w"); free(pth);
            if (!f[k]) rc = -3;
        }
        if (!rc && doDiffs) {
            fprintf(f[0], "# Input file:%s.vcf\n# Total number of segragating variant sites in this sample:%d\n# Richard's scoring scheme\n", noPath, totalVariantNumber);
            fprintf(f[1], "# Input file:%s.vcf\n# Total number of segragating variant sites in this sample: %d\n# Homozygous difference = 1, one homozygous, another heterozygous = 0.5\n", noPath, totalVariantNumber);
            fprintf(f[2], "# Input file:%s.vcf\n# number of sites both individuals hets/number of sites individuals have a homozygous difference; i.e. num(1/0::1/0)/num(1/1::0/0)\n# For a free mixing population, we expect this number ~2; for fully separated species ~0\n", noPath);
            fprintf(f[3], "# Input file:%s.vcf\n# Total number of segragating variant sites in this sample: %d\n# Pairwise missingness:\n", noPath, totalVariantNumber);
            for (int k = 0; k < 4; k++) for (int i = 0; i < n; i++) fprintf(f[k], "%s%s", names[i], i == n - 1 ? "\n" : "\t");
            print_matrix_d(f[0], diffs, n, 6);
            print_matrix_d(f[1], diffs_me, n, 10);                                  /* precision(10), :114 */
            print_matrix_d(f[2], hvh, n, 6);
            for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) fprintf(f[3], "%d%s", miss[(size_t)i * (size_t)n + (size_t)j], j == n - 1 ? "\n" : "\t");
        }
        for (int k = 0; k < 4; k++) if (f[k]) fclose(f[k]);
    }
    for (int b = 0; b < bootstrapBlockNum; b++) { free(bootBlocks[b]); free(bootMissBlocks[b]); }
    free(bootBlocks); free(bootMissBlocks);
    for (int i = 0; names && i < n; i++) free(names[i]);
    free(names); free(diffs); free(diffs_me); free(hvh); free(boot); free(miss); free(bootMiss); free(iv); free(hp); free(ms);
    free(hetCounts); free(hetsShared);
    free(fields); free(root);
    lr_close(&vcf);
    return rc;
}
