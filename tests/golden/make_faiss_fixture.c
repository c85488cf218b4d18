/* make_faiss_fixture.c -- test-fixture generator (NOT product code, not an importer).
 *
 * Writes two tiny Faiss index files by following, macro for macro, the serialisation sequence of
 * faiss/impl/index_write.cpp (Faiss 1.7.x): WRITE1 = fwrite of the raw little-endian object, WRITEVECTOR =
 * size_t count + raw elements, WRITEXBVECTOR = size_t (byte count / 4) + raw bytes.
 *   write_index_header : d (int) ntotal (int64) dummy dummy (int64 1<<20) is_trained (bool) metric_type (int)
 *   IndexFlatL2   "IxF2": header, WRITEXBVECTOR(codes)
 *   IndexIVFFlat  "IwFl": write_ivf_header = header, nlist (size_t), nprobe (size_t), write_index(quantizer),
 *                         write_direct_map = type (char) + WRITEVECTOR(array); then write_InvertedLists -- NO code_size field
 *   ArrayInvertedLists "ilar": nlist (size_t), code_size (size_t), "full" + WRITEVECTOR(sizes) (more than half the lists
 *                         non-empty) or "sprs" + WRITEVECTOR(list, size pairs); then per non-empty list codes, ids (int64)
 * It is written independently of obs_rvc_amd/faiss_index.py (no shared code) so that the reader is checked against a second
 * statement of the layout.   build + run:  gcc -O1 make_faiss_fixture.c -o /tmp/mk && /tmp/mk tests/golden
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define WRITE1(x) fwrite(&(x), sizeof(x), 1, f)
static void fourcc(FILE *f, const char *s) { fwrite(s, 1, 4, f); }
static void header(FILE *f, int d, int64_t ntotal)
{
    int64_t dummy = 1 << 20; unsigned char trained = 1; int metric = 1; /* METRIC_L2 */
    WRITE1(d); WRITE1(ntotal); WRITE1(dummy); WRITE1(dummy); WRITE1(trained); WRITE1(metric);
}
static void flat(FILE *f, int d, int64_t n, const float *x)
{
    fourcc(f, "IxF2"); header(f, d, n);
    size_t sz = (size_t)n * d * sizeof(float) / 4; WRITE1(sz); fwrite(x, 4, sz, f);
}
/* deterministic values: v[i][j] = (i * 7 + j * 3) % 11 - 5 + 0.25 * j */
static float val(int i, int j) { return (float)((i * 7 + j * 3) % 11 - 5) + 0.25f * (float)j; }

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : ".";
    char path[512];
    enum { D = 4, N = 7, NL = 3 };
    float x[N * D];
    for (int i = 0; i < N; i++) for (int j = 0; j < D; j++) x[i * D + j] = val(i, j);
    snprintf(path, sizeof path, "%s/faiss_flat.index", dir);
    FILE *f = fopen(path, "wb"); if (!f) return 1;
    flat(f, D, N, x); fclose(f);

    /* IVF3,Flat: list of id i = i % 3 for i < 6, id 6 -> list 0: sizes {3, 2, 2} -> "full"; a second file with only list 1
     * populated -> "sprs" */
    for (int sparse = 0; sparse < 2; sparse++) {
        snprintf(path, sizeof path, "%s/%s", dir, sparse ? "faiss_ivf_sparse.index" : "faiss_ivf.index");
        f = fopen(path, "wb"); if (!f) return 1;
        const int nl = sparse ? 5 : NL;
        int assign[N];
        for (int i = 0; i < N; i++) assign[i] = sparse ? 1 : (i < 6 ? i % 3 : 0);
        fourcc(f, "IwFl"); header(f, D, N);
        size_t nlist = (size_t)nl, nprobe = 1; WRITE1(nlist); WRITE1(nprobe);
        float cent[5 * D]; for (int c = 0; c < nl; c++) for (int j = 0; j < D; j++) cent[c * D + j] = (float)c - 0.5f * (float)j;
        flat(f, D, nl, cent);                                    /* quantizer */
        char dm = 0; WRITE1(dm); size_t zero = 0; WRITE1(zero); /* direct map: NoMap, empty array */
        fourcc(f, "ilar"); size_t cs = D * sizeof(float); WRITE1(nlist); WRITE1(cs);
        size_t sizes[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < N; i++) sizes[assign[i]]++;
        size_t non0 = 0; for (int c = 0; c < nl; c++) non0 += sizes[c] > 0;
        if (non0 > nlist / 2) { fourcc(f, "full"); WRITE1(nlist); fwrite(sizes, sizeof(size_t), nlist, f); }
        else {
            fourcc(f, "sprs"); size_t cnt = 2 * non0; WRITE1(cnt);
            for (int c = 0; c < nl; c++) if (sizes[c]) { size_t a = (size_t)c, b = sizes[c]; WRITE1(a); WRITE1(b); }
        }
        for (int c = 0; c < nl; c++) {
            if (!sizes[c]) continue;
            /* ids are stored in insertion order; insert in DEcreasing id order inside a list so storage order != id order */
            for (int i = N - 1; i >= 0; i--) if (assign[i] == c) fwrite(&x[i * D], sizeof(float), D, f);
            for (int i = N - 1; i >= 0; i--) if (assign[i] == c) { int64_t id = i; WRITE1(id); }
        }
        fclose(f);
    }
    return 0;
}
