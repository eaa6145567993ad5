"""Generator of tools/experiments/issue_probe.hip: what a wave pays for the instructions it issues beside exact-fp32 MFMAs.

The two F(4x4) Winograd loops (crdr_amd/csrc/wino4.hip, wino4_wgrad.hip) run ONE wave per SIMD (512 registers), so whatever that wave
issues between two `v_mfma_f32_16x16x4_f32` either hides in the MFMA's 32 cycles or costs matrix time.  Each probe kernel is one
inline-asm loop of MFMA slots (72 = the shape of a wino4 sub-step) with a chosen mix of fillers per slot -- VALU, LDS reads, LDS-DMA
pieces, stores -- timed with s_memtime; the program prints shader cycles per iteration for every mix (floor: 72 x 32 = 2304).

    python tools/experiments/issue_probe_gen.py > tools/experiments/issue_probe.hip
    hipcc -O2 --offload-arch=gfx950 tools/experiments/issue_probe.hip -o _exp/issue_probe && _exp/issue_probe   (on the GPU box)

Mix language ('+'-joined parts; a leading 'w2:' = two waves per SIMD with 36 MFMA slots each, 'vf:' = the VALU of a slot BEFORE its
memory-side fillers, 'nomfma:' = the fillers alone):
    mfma            nothing else
    valu<N>         N v_fma_f32 spread evenly over the slots        cv<N>x<G>   N v_fma_f32 in G evenly spaced clusters
    pk<N>x<G>       N v_pk_fma_f32 in G clusters                     ar<N>x<G>   N v_accvgpr_read_b32 in G clusters
    ldsf / ldsf128 / ldsf32 / ldsr   36 ds_read_b64 / 18 ds_read_b128 / 72 ds_read_b32 / 18 ds_read2st64_b32
    dma<N>          N LDS-DMA pieces (1 KiB each) spread evenly       dmag<N>   back to back
    st<N>           N buffer_store_dwordx4 (scattered 64-byte segments) spread evenly; sth / stq / stc: 128-byte / 256-byte segments / contiguous
"""
NSLOT = 72


def mfma(i):
    if i < 64:
        return f"v_mfma_f32_16x16x4_f32 a[{4 * i}:{4 * i + 3}], v1, v2, a[{4 * i}:{4 * i + 3}]"
    j = i - 64
    return f"v_mfma_f32_16x16x4_f32 v[{200 + 4 * j}:{203 + 4 * j}], v1, v2, v[{200 + 4 * j}:{203 + 4 * j}]"


class Body:
    def __init__(self, nslot=NSLOT, valu_first=False, nacc=72):
        self.nslot, self.nacc = nslot, nacc
        self.slots = [[] for _ in range(nslot)]    # memory-side fillers (LDS reads, DMA pieces, stores) of a slot
        self.vslots = [[] for _ in range(nslot)]   # VALU fillers of a slot
        self.valu_first = valu_first
        self.nv = self.nl = self.nd = 0

    def valu(self, s, n=1):
        for _ in range(n):
            r = 40 + self.nv % 32
            self.vslots[s].append(f"v_fma_f32 v{r}, v3, v4, v{r}")
            self.nv += 1

    def pk(self, s, n=1):
        for _ in range(n):
            r = 40 + 2 * (self.nv % 16)
            self.vslots[s].append(f"v_pk_fma_f32 v[{r}:{r + 1}], v[10:11], v[12:13], v[{r}:{r + 1}]")
            self.nv += 1

    def accread(self, s, n=1):
        for _ in range(n):
            self.vslots[s].append(f"v_accvgpr_read_b32 v{40 + self.nv % 32}, a{(self.nv * 7) % 128}")
            self.nv += 1

    def lds(self, s, kind):
        r = 80 + (self.nl * 4) % 32
        off = (self.nl * 1024) % 32768
        if kind == "b64":
            self.slots[s].append(f"ds_read_b64 v[{r}:{r + 1}], %2 offset:{off}")
        elif kind == "b128":
            self.slots[s].append(f"ds_read_b128 v[{r}:{r + 3}], %2 offset:{off}")
        elif kind == "b32":
            self.slots[s].append(f"ds_read_b32 v{r}, %2 offset:{off}")
        elif kind == "r2":
            self.slots[s].append(f"ds_read2st64_b32 v[{r}:{r + 1}], %2 offset0:{self.nl % 8} offset1:{16 + self.nl % 8}")
        self.nl += 1

    def dma(self, s):
        # one 1 KiB LDS-DMA piece: M0 = LDS destination (wave-uniform), per-lane source offset in %3, scalar offset walks an L2-resident buffer
        self.slots[s].append(f"s_add_u32 m0, %5, {0x10000 + (self.nd % 32) * 1024}")
        self.slots[s].append(f"s_add_u32 s40, %6, {(self.nd % 13) * 65536}")
        self.slots[s].append("buffer_load_dwordx4 %3, %4, s40 offen lds")
        self.nd += 1

    def store(self, s, kind=""):
        # 16 B per lane, the 16 lanes of a group 2 KiB apart, the four groups adjacent: 16 segments of 64 B (the epilogue's store pattern);
        # kind h / q / c: 8 segments of 128 B, 4 of 256 B, 1 KiB contiguous (what a lane permutation in front of the stores could buy)
        self.slots[s].append(f"s_add_u32 s40, %6, {(self.nd % 32) * 65536}")
        self.slots[s].append("buffer_store_dwordx4 v[14:17], %s, %%4, s40 offen" % {"": "%12", "h": "%13", "q": "%14", "c": "%15"}[kind])
        self.nd += 1

    def text(self):
        out = []
        for i in range(self.nslot):
            if self.nacc:
                out.append(mfma(i % self.nacc))
            out += (self.vslots[i] + self.slots[i]) if self.valu_first else (self.slots[i] + self.vslots[i])
        return out


def pattern(name):
    opts = name.split(":")
    flags, parts = opts[:-1], opts[-1].split("+")
    w2 = "w2" in flags
    ns = 36 if w2 else NSLOT
    b = Body(nslot=ns, valu_first="vf" in flags, nacc=0 if "nomfma" in flags else (36 if w2 else 72))

    def clusters(p):
        n, g = p.split("x")
        return int(n), int(g)
    for p in parts:
        if p == "mfma":
            pass
        elif p.startswith("valu"):
            k = int(p[4:])
            for i in range(k):
                b.valu((i * ns) // k)
        elif p.startswith("cv"):
            n, g = clusters(p[2:])
            for c in range(g):
                b.valu(min(ns - 1, (c * ns) // g + 1), n // g + (1 if c < n % g else 0))
        elif p.startswith("pk"):
            n, g = clusters(p[2:])
            for c in range(g):
                b.pk(min(ns - 1, (c * ns) // g + 1), n // g + (1 if c < n % g else 0))
        elif p.startswith("ar"):
            n, g = clusters(p[2:])
            for c in range(g):
                b.accread(min(ns - 1, (c * ns) // g + 1), n // g + (1 if c < n % g else 0))
        elif p == "ldsf":
            for j in range(ns // 2):
                b.lds(2 * j + 1, "b64")
        elif p == "ldsf128":
            for j in range(ns // 4):
                b.lds(4 * j + 1, "b128")
        elif p == "ldsf32":
            for j in range(ns):
                b.lds(j, "b32")
        elif p == "ldsr":
            for j in range(ns // 4):
                b.lds(2 * j, "r2")
        elif p.startswith("dmag"):
            for q in range(int(p[4:])):
                b.dma(min(ns - 1, ns // 2 + q))
        elif p.startswith("dma"):
            n = int(p[3:])
            for q in range(n):
                b.dma(min(ns - 1, (q * ns) // n + 2))
        elif p.startswith("st"):
            kind = p[2] if p[2] in "hqc" else ""
            n = int(p[3:] if kind else p[2:])
            for q in range(n):
                b.store((q * ns) // n, kind)
        else:
            raise SystemExit("unknown part " + p)
    return b, w2


PATTERNS = [
    "mfma", "valu72", "valu144",
    # does the order inside a gap matter?  memory-side fillers right behind the MFMA (default) or behind the VALU (vf:)
    "valu144+ldsf+ldsr+dma13", "vf:valu144+ldsf+ldsr+dma13",
    # the same 144 / 200 VALU in fewer, larger clusters
    "cv144x72", "cv144x36", "cv144x18", "cv144x9", "cv144x4", "cv144x1",
    "cv200x18+ldsf+ldsr+dma13", "cv200x9+ldsf+ldsr+dma13", "cv144x9+ldsf+ldsr+dma13",
    # the position-split form: half the transform, 16-byte filter reads
    "cv92x12+ldsf128+ldsr+dma13", "cv92x6+ldsf128+ldsr+dma13", "cv72x6+ldsf128+ldsr+dma13", "valu92+ldsf128+ldsr+dma13",
    # packed fp32, accumulator reads, stores
    "pk72x72", "pk72x9", "ar144x9", "ar288x1", "st32", "sth32", "stq32", "stc32", "nomfma:sth32", "nomfma:stc32", "cv144x9+st32", "nomfma:st32", "nomfma:cv144x1", "nomfma:ar288x1", "nomfma:ldsf+ldsr",
    "nomfma:dma13", "nomfma:dma52",
    # two waves per SIMD, 36 MFMA slots each: does one wave's VALU hide under the other's MFMAs?
    # the current one-wave loop's mix and the two-wave form of it (per wave: half the positions = 36 MFMAs, half the transform)
    "pk72x9+ldsf+ldsr+dma13", "pk78x10+ldsf+ldsr+dma13", "pk36x5+ldsf128+ldsr+dma13",
    "w2:pk36x5+ldsf+ldsr+dma7", "w2:pk36x3+ldsf+ldsr+dma7", "w2:pk36x9+ldsf+ldsr+dma7", "w2:pk36x5", "w2:pk72x5+ldsf+ldsr+dma7", "w2:pk18x3+ldsf+ldsr+dma7",
    "w2:mfma", "w2:valu36", "w2:valu72", "w2:valu144", "w2:cv72x4", "w2:cv144x4", "w2:valu72+ldsf+ldsr+dma13", "w2:cv144x4+ldsf+ldsr+dma13",
]


def clobbers(w2):
    c = [f"\"a{i}\"" for i in range(144 if w2 else 256)] + [f"\"v{i}\"" for i in range(40, 112 if w2 else 232)]
    c += ["\"s40\"", "\"s41\"", "\"s42\"", "\"m0\"", "\"scc\"", "\"memory\""]
    return ", ".join(c)


def kernel(idx, name):
    b, w2 = pattern(name)
    body = b.text()
    nd = b.nd
    init = [f"v_accvgpr_write_b32 a{i}, 0" for i in range(144 if w2 else 256)] + [f"v_mov_b32 v{i}, 0" for i in range(40, 112 if w2 else 232)]
    init += ["v_mov_b32 v10, %10", "v_mov_b32 v11, %10", "v_mov_b32 v12, %11", "v_mov_b32 v13, %11"] + [f"v_mov_b32 v{i}, %8" for i in range(14, 18)]
    lines = init + ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier", "s_memtime %0", "s_mov_b32 s41, %7", f"LOOP{idx}_%=:"] + body
    lines += [f"s_waitcnt vmcnt({min(nd, 63)}) lgkmcnt(0)" if nd else "s_waitcnt lgkmcnt(0)", "s_barrier", "s_sub_u32 s41, s41, 1", "s_cmp_lg_u32 s41, 0",
              f"s_cbranch_scc1 LOOP{idx}_%=", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_memtime %1", "s_waitcnt lgkmcnt(0)"]
    asm = "\n".join(f"      \"{l}\\n\"" for l in lines)
    nt = 512 if w2 else 256
    has_store = any("buffer_store" in l for l in body)
    return f"""
__global__ __launch_bounds__({nt}) void probe{idx}(float* out, unsigned long long* cyc, const float* gsrc, float* gdst, int iters) {{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 40 * 1024; i += {nt}) smem[i] = (float)(i & 15) * 0.01f;
  __syncthreads();
  float a = 0.5f + lane * 0.001f, b = 0.25f - lane * 0.002f, c = 1.0001f, d = 0.0001f;
  unsigned ldsoff = (unsigned)(lane * 16 + (wave & 3) * 1024);
  unsigned goff = (unsigned)(lane * 16 + (wave & 3) * 1024 + (blockIdx.x & 31) * 4096);
  unsigned soff = (unsigned)((lane >> 4) * 16 + (lane & 15) * 2048 + wave * 64 + (blockIdx.x & 31) * (1 << 21));
  unsigned soffh = (unsigned)((lane & 7) * 16 + (lane >> 3) * 2048 + wave * 128 + (blockIdx.x & 31) * (1 << 21));
  unsigned soffq = (unsigned)((lane & 15) * 16 + (lane >> 4) * 2048 + wave * 256 + (blockIdx.x & 31) * (1 << 21));
  unsigned soffc = (unsigned)(lane * 16 + wave * 1024 + (blockIdx.x & 31) * (1 << 21));
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gsrc), 0, 1u << 22, 0x00020000);
  __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(gdst, 0, 1u << 27, 0x00020000);
  (void)rd; (void)rs;
  unsigned ldsbase = (unsigned)__builtin_amdgcn_readfirstlane((int)((wave & 3) * 32768 / 4));
  unsigned sbase = 0;
  unsigned long long t0, t1;
  asm volatile("v_mov_b32 v1, %8\\n v_mov_b32 v2, %9\\n v_mov_b32 v3, %10\\n v_mov_b32 v4, %11\\n"
{asm}
      : "=&s"(t0), "=&s"(t1)
      : "v"(ldsoff), "v"(goff), "s"({'rd' if has_store else 'rs'}), "s"(ldsbase), "s"(sbase), "s"(iters), "v"(a), "v"(b), "v"(c), "v"(d), "v"(soff), "v"(soffh), "v"(soffq), "v"(soffc)
      : "v1", "v2", "v3", "v4", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", {clobbers(w2)});
  if (lane == 0) cyc[blockIdx.x * {nt // 64} + wave] = t1 - t0;
  if (out) out[threadIdx.x] = a;
}}
"""


def main():
    print("// GENERATED by tools/experiments/issue_probe_gen.py -- do not edit")
    print("#include <hip/hip_runtime.h>\n#include <algorithm>\n#include <cstdio>\n#include <vector>\n")
    w2s = []
    for i, n in enumerate(PATTERNS):
        print(kernel(i, n))
        w2s.append("1" if "w2" in n.split(":")[:-1] else "0")
    print("typedef void (*probe_fn)(float*, unsigned long long*, const float*, float*, int);")
    print("static probe_fn fns[] = {" + ", ".join(f"probe{i}" for i in range(len(PATTERNS))) + "};")
    print("static const char* names[] = {" + ", ".join(f"\"{n}\"" for n in PATTERNS) + "};")
    print("static const int two[] = {" + ", ".join(w2s) + "};")
    print(r"""
int main() {
  const int nblk = 256, iters = 2000;
  unsigned long long* cyc; float* gsrc; float* gdst;
  (void)hipMalloc(&cyc, nblk * 8 * 8); (void)hipMalloc(&gsrc, 8 << 20); (void)hipMemset(gsrc, 0, 8 << 20); (void)hipMalloc(&gdst, 1 << 27);
  std::vector<unsigned long long> h(nblk * 8);
  printf("%-40s %10s %10s %10s  %8s\n", "mix (per 72 MFMA slots per SIMD)", "cyc/iter", "p90", "excess", "us/iter");
  for (size_t k = 0; k < sizeof(fns) / sizeof(fns[0]); ++k) {
    const int nt = two[k] ? 512 : 256, nw = nt / 64;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fns[k]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fns[k], dim3(nblk), dim3(nt), 160 * 1024, 0, nullptr, cyc, gsrc, gdst, 200);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fns[k], dim3(nblk), dim3(nt), 160 * 1024, 0, nullptr, cyc, gsrc, gdst, iters);
    (void)hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", names[k]); return 1; }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h.data(), cyc, nblk * nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + nblk * nw);
    const double med = (double)h[nblk * nw / 2] / iters, p90 = (double)h[nblk * nw * 9 / 10] / iters;
    printf("%-40s %10.1f %10.1f %10.1f  %8.3f\n", names[k], med, p90, med - 2304.0, ms * 1e3 / iters);
  }
  return 0;
}""")


if __name__ == "__main__":
    main()
