// Database build entry points: PQ encode (quantizers.hpp:222-245), the compute of
// index_db::add_vectors (databases.hpp:270-298) and the k-means iterations (databases.cpp:50-90).  Stateless: host buffers
// (or device pointers) in and out, any device.
#include "qadc_host.h"

using namespace qadc;
using namespace qadc::host;

extern "C" {

int qadc_pq_encode(int M, int dim, const float* codebooks, const void* d_vectors, uint64_t n, void* d_codes, int device_id) {
    return qadc_pq_encode_mode(M, dim, codebooks, d_vectors, n, d_codes, 1, 1, device_id);
}

int qadc_pq_encode_mode(int M, int dim, const float* codebooks, const void* d_vectors, uint64_t n, void* d_codes, int encode_form,
                        int sum_mode, int device_id) {
    if ((M != 16 && M != 32) || dim <= 0 || dim % M != 0 || !codebooks || (n && (!d_vectors || !d_codes)))
        return fail(QADC_E_ARG, "bad arguments");
    HIPCHECK(hipSetDevice(device_id));
    const size_t ncb = (size_t)M * 16 * (dim / M);
    float* d_cb = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_cb), ncb * sizeof(float)));
    HIPCHECK(hipMemcpy(d_cb, codebooks, ncb * sizeof(float), hipMemcpyHostToDevice));
    if (n) launch_pq_encode(static_cast<const float*>(d_vectors), n, M, dim, d_cb, encode_form != 0, sum_mode != 0,
                            static_cast<uint8_t*>(d_codes), nullptr);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipFree(d_cb));
    return QADC_OK;
}

int qadc_pq_encode_host(int M, int dim, const float* codebooks, const float* vectors, uint64_t n, uint8_t* codes, int device_id) {
    return qadc_pq_encode_host_mode(M, dim, codebooks, vectors, n, codes, 1, 1, device_id);
}

int qadc_pq_encode_host_mode(int M, int dim, const float* codebooks, const float* vectors, uint64_t n, uint8_t* codes, int encode_form,
                             int sum_mode, int device_id) {
    if (!vectors || !codes) return fail(QADC_E_ARG, "bad arguments");
    HIPCHECK(hipSetDevice(device_id));
    float* d_v = nullptr;
    uint8_t* d_c = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_v), std::max<size_t>(1, n * dim * sizeof(float))));
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_c), std::max<size_t>(1, n * (M / 2))));
    HIPCHECK(hipMemcpy(d_v, vectors, n * dim * sizeof(float), hipMemcpyHostToDevice));
    const int rc = qadc_pq_encode_mode(M, dim, codebooks, d_v, n, d_c, encode_form, sum_mode, device_id);
    if (rc == QADC_OK) HIPCHECK(hipMemcpy(codes, d_c, n * (M / 2), hipMemcpyDeviceToHost));
    HIPCHECK(hipFree(d_v));
    HIPCHECK(hipFree(d_c));
    return rc;
}

/* ---- database build (N4): index_db::add_vectors' compute and the k-means iterations, host buffers in and out ---- */
extern "C++" {
namespace {
struct ScratchFree {                                           // hipFree on every exit path
    std::vector<void*> p;
    ~ScratchFree() { for (void* x : p) if (x) (void)hipFree(x); }
    template <typename T> hipError_t alloc(T** out, size_t bytes) {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) { p.push_back(q); *out = static_cast<T*>(q); }
        return e;
    }
};
constexpr uint64_t kBuildChunk = 32768;                        // vectors per pass (the distance scratch is chunk x K floats)

// nearest centroid of every vector, chunk by chunk: the coarse kernels of qadc_search with ma = 1 (find_k_neighbors with k = 1 on the
// expansion distances).  d_dist: chunk x K distances + chunk query norms + K centroid norms (the centroids may have moved: each call).
int assign_nearest(const float* d_vectors, uint64_t n, int dim, int K, const float* d_coarse, float* d_dist, int32_t* d_assign, int sum_mode) {
    const uint64_t chunk = std::min<uint64_t>(kBuildChunk, std::max<uint64_t>(n, 1));
    float* d_qnorm = d_dist + chunk * (uint64_t)K;
    float* d_cnorm = d_qnorm + chunk;
    launch_row_sqnorm(d_coarse, K, dim, sum_mode, d_cnorm, nullptr);
    for (uint64_t o = 0; o < n; o += kBuildChunk) {
        const int cnt = (int)std::min<uint64_t>(kBuildChunk, n - o);
        launch_coarse_assign(d_vectors + o * dim, d_coarse, cnt, K, dim, 1, d_qnorm, d_cnorm, sum_mode, d_dist, d_assign + o, nullptr);
    }
    HIPCHECK(hipGetLastError());
    return QADC_OK;
}
}  // namespace
}  // extern "C++"

int qadc_ivf_encode_host(int M, int dim, const float* codebooks, const float* rotation, int K, const float* coarse,
                         const float* vectors, uint64_t n, int32_t* assign_out, uint8_t* codes, int device_id) {
    return qadc_ivf_encode_host_mode(M, dim, codebooks, rotation, K, coarse, vectors, n, assign_out, codes, 1, 1, device_id);
}

int qadc_ivf_encode_host_mode(int M, int dim, const float* codebooks, const float* rotation, int K, const float* coarse,
                              const float* vectors, uint64_t n, int32_t* assign_out, uint8_t* codes, int encode_form, int sum_mode,
                              int device_id) {
    if ((M != 16 && M != 32) || dim <= 0 || dim % M != 0 || !codebooks || K < 0 || (K > 0 && !coarse) ||
        (n && (!vectors || !codes)))
        return fail(QADC_E_ARG, "bad arguments");
    HIPCHECK(hipSetDevice(device_id));
    ScratchFree mem;
    const size_t ncb = (size_t)M * 16 * (dim / M);
    float *d_cb = nullptr, *d_rot = nullptr, *d_coarse = nullptr, *d_v = nullptr, *d_x = nullptr, *d_dist = nullptr;
    int32_t* d_assign = nullptr;
    uint8_t* d_codes = nullptr;
    HIPCHECK(mem.alloc(&d_cb, ncb * sizeof(float)));
    HIPCHECK(hipMemcpy(d_cb, codebooks, ncb * sizeof(float), hipMemcpyHostToDevice));
    if (rotation) {
        HIPCHECK(mem.alloc(&d_rot, sizeof(float) * (size_t)dim * dim));
        HIPCHECK(hipMemcpy(d_rot, rotation, sizeof(float) * (size_t)dim * dim, hipMemcpyHostToDevice));
    }
    if (K > 0) {
        HIPCHECK(mem.alloc(&d_coarse, sizeof(float) * (size_t)K * dim));
        HIPCHECK(hipMemcpy(d_coarse, coarse, sizeof(float) * (size_t)K * dim, hipMemcpyHostToDevice));
        HIPCHECK(mem.alloc(&d_dist, sizeof(float) * ((size_t)std::min<uint64_t>(kBuildChunk, std::max<uint64_t>(n, 1)) * (K + 1) + K)));
        HIPCHECK(mem.alloc(&d_assign, sizeof(int32_t) * n));
    }
    HIPCHECK(mem.alloc(&d_v, sizeof(float) * n * dim));
    HIPCHECK(mem.alloc(&d_codes, n * (size_t)(M / 2)));
    HIPCHECK(hipMemcpy(d_v, vectors, sizeof(float) * n * dim, hipMemcpyHostToDevice));
    const float* d_enc = d_v;
    if (n && K > 0)
        if (int rc = assign_nearest(d_v, n, dim, K, d_coarse, d_dist, d_assign, sum_mode != 0)) return rc;
    if (n && (K > 0 || rotation)) {
        HIPCHECK(mem.alloc(&d_x, sizeof(float) * n * dim));
        launch_residual_rotate(d_v, n, dim, d_coarse, d_assign, d_rot, d_x, nullptr);
        d_enc = d_x;
    }
    if (n) launch_pq_encode(d_enc, n, M, dim, d_cb, encode_form != 0, sum_mode != 0, d_codes, nullptr);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(codes, d_codes, n * (size_t)(M / 2), hipMemcpyDeviceToHost));
    if (assign_out && K > 0) HIPCHECK(hipMemcpy(assign_out, d_assign, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return QADC_OK;
}

int qadc_kmeans_iterations_host(const float* vectors, uint64_t n, int dim, int K, float* centroids, int iters, int32_t* assign_out,
                                int device_id) {
    return qadc_kmeans_iterations_host_mode(vectors, n, dim, K, centroids, iters, assign_out, 1, device_id);
}

int qadc_kmeans_iterations_host_mode(const float* vectors, uint64_t n, int dim, int K, float* centroids, int iters, int32_t* assign_out,
                                     int div_mode, int device_id) {
    if (!vectors || !centroids || n == 0 || dim <= 0 || dim > 2048 || K <= 0 || iters < 0) return fail(QADC_E_ARG, "bad arguments");
    HIPCHECK(hipSetDevice(device_id));
    ScratchFree mem;
    float *d_v = nullptr, *d_c = nullptr, *d_dist = nullptr;
    int32_t* d_assign = nullptr;
    HIPCHECK(mem.alloc(&d_v, sizeof(float) * n * dim));
    HIPCHECK(mem.alloc(&d_c, sizeof(float) * (size_t)K * dim));
    HIPCHECK(mem.alloc(&d_dist, sizeof(float) * ((size_t)std::min<uint64_t>(kBuildChunk, n) * (K + 1) + K)));
    HIPCHECK(mem.alloc(&d_assign, sizeof(int32_t) * n));
    HIPCHECK(hipMemcpy(d_v, vectors, sizeof(float) * n * dim, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_c, centroids, sizeof(float) * (size_t)K * dim, hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(d_assign, 0, sizeof(int32_t) * n));
    for (int it = 0; it < iters; ++it) {                       // databases.cpp:57-89
        if (int rc = assign_nearest(d_v, n, dim, K, d_c, d_dist, d_assign, 1)) return rc;
        launch_kmeans_update(d_v, n, dim, K, d_assign, d_c, div_mode != 0, nullptr);
    }
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(centroids, d_c, sizeof(float) * (size_t)K * dim, hipMemcpyDeviceToHost));
    if (assign_out) HIPCHECK(hipMemcpy(assign_out, d_assign, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return QADC_OK;
}

}  // extern "C"
