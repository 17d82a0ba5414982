/* Oracle (TEST INFRASTRUCTURE ONLY): C restatement of the in-database form of the search,
 *     ORDER BY e.embedding <#> q ASC LIMIT k            (streamlit_app.py:282-283)
 * pgvector is a third-party Postgres extension that is not part of /root/reference (requirements.txt:2,
 * unpinned).  Its published algorithm for "<#>" on vector(d) is the negative inner product accumulated
 * in float over the dimensions in index order, evaluated row by row in a sequential scan (the schema
 * declares no ANN index, rds_schema.sql:41-56); ORDER BY .. LIMIT k keeps the k smallest distances
 * (top-N heapsort).  Ties are broken here by row id ascending (Postgres' own tie order is unspecified).
 *
 *   distance[i] = -(sum_j (float)e[i][j] * q[j])   accumulated in fp32, j = 0..d-1
 *
 * Built by oracle/Makefile into oracle/_build/libpgvector_ip.so; only tests/ load it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct {
    float dist;
    int64_t row;
} item;

/* a is "worse" than b when it would be returned later: larger distance, then larger row */
static int worse(item a, item b) { return a.dist > b.dist || (a.dist == b.dist && a.row > b.row); }

static void sift_down(item *h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && worse(h[l], h[m])) m = l;
        if (r < n && worse(h[r], h[m])) m = r;
        if (m == i) return;
        item t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

static int cmp_items(const void *x, const void *y) {
    item a = *(const item *)x, b = *(const item *)y;
    if (worse(a, b)) return 1;
    if (worse(b, a)) return -1;
    return 0;
}

float pgv_neg_inner_product(const float *a, const float *b, int d) {
    float acc = 0.0f;
    for (int j = 0; j < d; ++j) acc += a[j] * b[j];
    return -acc;
}

/* Fills out_rows / out_dist (k entries; padded with row -1, dist +inf). NaN distances are skipped. */
int pgv_order_by_ip_limit(const float *emb, int64_t n, int d, const float *q, int k, int64_t *out_rows, float *out_dist) {
    if (k <= 0) return -1;
    item *heap = (item *)malloc(sizeof(item) * (size_t)k);
    if (!heap) return -2;
    int cnt = 0;
    for (int64_t i = 0; i < n; ++i) {
        item it = {pgv_neg_inner_product(emb + i * d, q, d), i};
        if (it.dist != it.dist) continue;
        if (cnt < k) {
            heap[cnt++] = it;
            if (cnt == k)
                for (int s = k / 2 - 1; s >= 0; --s) sift_down(heap, k, s);
        } else if (worse(heap[0], it)) {
            heap[0] = it;
            sift_down(heap, k, 0);
        }
    }
    qsort(heap, (size_t)cnt, sizeof(item), cmp_items);
    for (int i = 0; i < k; ++i) {
        out_rows[i] = i < cnt ? heap[i].row : -1;
        out_dist[i] = i < cnt ? heap[i].dist : INFINITY;
    }
    free(heap);
    return 0;
}
