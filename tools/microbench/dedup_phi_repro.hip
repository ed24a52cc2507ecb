// dedup_phi_repro.hip -- stand-alone check of the claim at tapir_amd/csrc/pattern_kernels.hpp (dedup_resolve_kernel):
// "the hipcc of ROCm 7.2 lost the representative in the phi of `rep = same ? r : -1` after the comparison loop".
// Two forms of the same kernel on the same inputs: `phi_form` merges the result into one value stored once after the
// branch (the form the product kernel had), `store_form` stores on each path (the form it has now).  Both are compared
// with the host's answer; exit code 1 and a count when a form is wrong.  Re-run on a new ROCm to see whether the
// workaround is still needed.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/dedup_phi_repro tools/microbench/dedup_phi_repro.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

struct P {
    const int64_t* off; const int32_t* chunk_locus; const int32_t* chunk_index; const uint8_t* on; uint8_t* flag;
    const unsigned long long* hash; const unsigned long long* tab_key; const int32_t* tab_val; const uint32_t* packed;
    int32_t nwords; int64_t ncols; int32_t* dup_of;
};
constexpr uint8_t kDup = 9;
__device__ inline unsigned long long key(unsigned long long h) { return h | 1ull; }

__global__ __launch_bounds__(256) void phi_form(P p) {
    const int locus = p.chunk_locus[blockIdx.x];
    const int64_t lo = p.off[locus], hi = p.off[locus + 1];
    const int64_t base = lo + (int64_t)p.chunk_index[blockIdx.x] * 1024;
    const bool on = p.on[locus] != 0;
    const unsigned long long size = 2ull * (unsigned long long)(hi - lo);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t c = base + j * 256 + threadIdx.x;
        if (c >= hi) continue;
        int32_t rep = -1;
        if (on && p.flag[c] == 0) {
            const unsigned long long h = key(p.hash[c]);
            unsigned long long slot = h % size;
            while (p.tab_key[2 * lo + slot] != h) slot = (slot + 1 == size) ? 0 : slot + 1;
            const int64_t r = lo + p.tab_val[2 * lo + slot];
            bool same = r != c;
            for (int w = 0; w < p.nwords && same; ++w) same = p.packed[(int64_t)w * p.ncols + c] == p.packed[(int64_t)w * p.ncols + r];
            rep = same ? (int32_t)r : -1;
            if (same) p.flag[c] = kDup;
        }
        p.dup_of[c] = rep;
    }
}

__global__ __launch_bounds__(256) void store_form(P p) {
    const int locus = p.chunk_locus[blockIdx.x];
    const int64_t lo = p.off[locus], hi = p.off[locus + 1];
    const int64_t base = lo + (int64_t)p.chunk_index[blockIdx.x] * 1024;
    const bool on = p.on[locus] != 0;
    const unsigned long long size = 2ull * (unsigned long long)(hi - lo);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t c = base + j * 256 + threadIdx.x;
        if (c >= hi) continue;
        p.dup_of[c] = -1;
        if (on && p.flag[c] == 0) {
            const unsigned long long h = key(p.hash[c]);
            unsigned long long slot = h % size;
            while (p.tab_key[2 * lo + slot] != h) slot = (slot + 1 == size) ? 0 : slot + 1;
            const int64_t r = lo + p.tab_val[2 * lo + slot];
            unsigned differ = (r == c) ? 1u : 0u;
            for (int w = 0; w < p.nwords; ++w) differ |= p.packed[(int64_t)w * p.ncols + c] ^ p.packed[(int64_t)w * p.ncols + r];
            if (differ == 0u) { p.dup_of[c] = (int32_t)r; p.flag[c] = kDup; }
        }
    }
}

template <typename T> T* up(const std::vector<T>& h) {
    T* d; hipMalloc((void**)&d, h.size() * sizeof(T)); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d;
}

int main() {
    const int L = 3, n = 3000, nw = 8;               // three loci of 3000 columns; every column repeats one of 40 patterns
    const int64_t N = (int64_t)L * n;
    std::vector<int64_t> off = {0, n, 2 * n, 3 * n};
    std::vector<uint32_t> packed((size_t)nw * N);
    std::vector<unsigned long long> hash(N), tab_key(2 * N, ~0ull);
    std::vector<int32_t> tab_val(2 * N, 0x7fffffff), want(N, -1), chunk_locus, chunk_index;
    std::vector<uint8_t> flag(N, 0), on = {1, 0, 1};
    uint64_t s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
    for (int l = 0; l < L; ++l)
        for (int c = 0; c < n; ++c) {
            const int64_t g = off[l] + c;
            const uint32_t pat = rnd() % 40;
            for (int w = 0; w < nw; ++w) packed[(size_t)w * N + g] = pat * 2654435761u + w * 97u + l;
            hash[g] = (unsigned long long)pat * 0x9E3779B97F4A7C15ull + l;
            if (c % 17 == 0) flag[g] = 3;            // columns the optimiser never sees
        }
    for (int l = 0; l < L; ++l) {                    // host twin of dedup_insert_kernel: first column of each key
        const unsigned long long size = 2ull * n;
        for (int c = 0; c < n; ++c) {
            const int64_t g = off[l] + c;
            if (flag[g]) continue;
            const unsigned long long h = hash[g] | 1ull;
            unsigned long long slot = h % size;
            while (tab_key[2 * off[l] + slot] != ~0ull && tab_key[2 * off[l] + slot] != h) slot = (slot + 1 == size) ? 0 : slot + 1;
            tab_key[2 * off[l] + slot] = h;
            if (c < tab_val[2 * off[l] + slot]) tab_val[2 * off[l] + slot] = c;
        }
        for (int c = 0; c < n; ++c) {
            const int64_t g = off[l] + c;
            if (flag[g] || !on[l]) continue;
            const unsigned long long h = hash[g] | 1ull;
            unsigned long long slot = h % size;
            while (tab_key[2 * off[l] + slot] != h) slot = (slot + 1 == size) ? 0 : slot + 1;
            const int64_t r = off[l] + tab_val[2 * off[l] + slot];
            if (r != g) want[g] = (int32_t)r;
        }
        for (int k = 0; k * 1024 < n; ++k) { chunk_locus.push_back(l); chunk_index.push_back(k); }
    }
    int bad_total = 0;
    for (int form = 0; form < 2; ++form) {
        P p;
        p.off = up(off); p.chunk_locus = up(chunk_locus); p.chunk_index = up(chunk_index); p.on = up(on); p.flag = up(flag);
        p.hash = up(hash); p.tab_key = up(tab_key); p.tab_val = up(tab_val); p.packed = up(packed); p.nwords = nw; p.ncols = N;
        std::vector<int32_t> got(N, -7);
        p.dup_of = up(got);
        if (form == 0) phi_form<<<dim3((unsigned)chunk_locus.size()), dim3(256)>>>(p);
        else store_form<<<dim3((unsigned)chunk_locus.size()), dim3(256)>>>(p);
        hipDeviceSynchronize();
        hipMemcpy(got.data(), p.dup_of, N * sizeof(int32_t), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int64_t g = 0; g < N; ++g) bad += got[g] != want[g];
        printf("%s: %d of %lld representatives wrong\n", form == 0 ? "phi_form (one merged store)" : "store_form (a store on each path)", bad, (long long)N);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
