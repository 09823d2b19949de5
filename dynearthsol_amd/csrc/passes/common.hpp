// passes/common.hpp -- Element material data shared by the passes: per-element marker mix, property means, k_props / k_ptab.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// =====================================================================================
// kernels
// =====================================================================================
struct ElemProps { double bulkm, shearm, phi, cp, k; };


// Instrumented builds only (tools/build_variant.sh NAME -DDES_STAMPS; tools/patch_phase_timing.py): the first lane of a
// wavefront of every workgroup of the patch passes stamps the 100-MHz wall clock at its phase boundaries into a buffer
// of the code object, which the tool reads after the run (the last launch of each pass wins).  The shipped library
// holds none of this.
#ifdef DES_STAMPS
#define DES_STAMP_WG 8192
#define DES_STAMP_SLOTS 8
__device__ unsigned long long g_stamps[4][DES_STAMP_SLOTS][DES_STAMP_WG];          // [EN1 | EN3 | E2 pipelined: per wavefront, summed over its tiles][slot][workgroup]
#define DES_STAMP(pass, slot) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < DES_STAMP_WG) g_stamps[pass][slot][blockIdx.x] = wall_clock64(); } while (0)
#define DES_STAMP0(pass, slot) do { if (threadIdx.x == 0 && blockIdx.x < DES_STAMP_WG) g_stamps[pass][slot][blockIdx.x] = wall_clock64(); } while (0)
#else
#define DES_STAMP(pass, slot) do {} while (0)
#define DES_STAMP0(pass, slot) do {} while (0)
#endif

// Element e of plane i of an SoA array [planes][ne], addressed as a UNIFORM plane base (a scalar register
// pair) + a 32-bit byte offset eo = 8 e that every plane shares: the lane holds one offset register instead
// of a 64-bit address per plane it loads and later stores (18 planes in the stress update = 36 VGPRs).
// ne < 2^29 (des_dev_create), so 8 e fits.
#define DES_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ double pl_ld(const double *__restrict__ a, int i, int ne, unsigned eo)
{
    const DES_GLOBAL char *b = (const DES_GLOBAL char *)(a + (size_t)i * ne);
    asm("" : "+s"(b));                 // the base stays a scalar pair of its own (not folded into a per-lane address chain)
    return *(const DES_GLOBAL double *)(b + eo);
}
// the same, non-temporal: planes that are read once and written once per step (nothing reads them again before
// ~0.6 GB of other traffic has passed) should not evict what the next pass is about to read
__device__ __forceinline__ double pl_ld_nt(const double *__restrict__ a, int i, int ne, unsigned eo)
{
    const DES_GLOBAL char *b = (const DES_GLOBAL char *)(a + (size_t)i * ne);
    asm("" : "+s"(b));
    return __builtin_nontemporal_load((const DES_GLOBAL double *)(b + eo));
}
__device__ __forceinline__ void pl_st_nt(double *__restrict__ a, int i, int ne, unsigned eo, double v)
{
    DES_GLOBAL char *b = (DES_GLOBAL char *)(a + (size_t)i * ne);
    asm("" : "+s"(b));
    __builtin_nontemporal_store(v, (DES_GLOBAL double *)(b + eo));
}
// record `id` of a nodal / element array addressed the same way (gathers: uniform base + 32-bit byte
// offset; nn < 2^27, des_dev_create)
typedef double des_dv4 __attribute__((ext_vector_type(4)));
typedef int des_iv4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rec_ld(const double *__restrict__ a, int id)
{
    const DES_GLOBAL char *b = (const DES_GLOBAL char *)a;
    asm("" : "+s"(b));
    return *(const DES_GLOBAL double *)(b + (unsigned)id * 8u);
}
__device__ __forceinline__ d4 rec_ld(const d4 *__restrict__ a, int id)
{
    const DES_GLOBAL char *b = (const DES_GLOBAL char *)a;
    asm("" : "+s"(b));
    const des_dv4 v = *(const DES_GLOBAL des_dv4 *)(b + (unsigned)id * 32u);
    d4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
}
__device__ __forceinline__ int4 rec_ld(const int4 *__restrict__ a, int id)
{
    const DES_GLOBAL char *b = (const DES_GLOBAL char *)a;
    asm("" : "+s"(b));
    const des_iv4 v = *(const DES_GLOBAL des_iv4 *)(b + (unsigned)id * 16u);
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void pl_st(double *__restrict__ a, int i, int ne, unsigned eo, double v)
{
    DES_GLOBAL char *b = (DES_GLOBAL char *)(a + (size_t)i * ne);
    asm("" : "+s"(b));
    *(DES_GLOBAL double *)(b + eo) = v;
}

// One listed patch element as EN1 / EN3 read it: engine/patch.hpp packs (element | owner flag, four local node ids,
// four CSR slots) into 16 bytes -- .x = elem (31 bits) | ln0 << 31 | ln1 << 40 | ln2 << 49, .y = ln3 | slot_k << (9 + 12 k),
// 0xfff = no slot.
struct PatchElem { int ew; unsigned short ln[4]; short sl[4]; };
__device__ __forceinline__ PatchElem patch_elem_unpack(const ulonglong2 r)
{
    PatchElem E;
    E.ew = (int)(r.x & 0x7fffffffull);
    E.ln[0] = (unsigned short)((r.x >> 31) & 0x1ff); E.ln[1] = (unsigned short)((r.x >> 40) & 0x1ff);
    E.ln[2] = (unsigned short)((r.x >> 49) & 0x1ff); E.ln[3] = (unsigned short)(r.y & 0x1ff);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int s = (int)((r.y >> (9 + 12 * k)) & 0xfff);
        E.sl[k] = (short)(s == 0xfff ? -1 : s);
    }
    return E;
}


// What the kernels know about the materials of an element (refresh_elem_cache,
// matprops.cxx:259-303, redone whenever the marker counts change):
//   markers [ne][nmat]  the counts themselves
//   mono    [ne]        (material << 16) | count where one material holds every marker, else -1:
//                       4 bytes per element and pass instead of 4*nmat + 40
//   props   [5][ne]     bulkm, shearm, phi, cp, k of every element (nmat > 1 only)
//   ptab    [nmat][DES_PTAB_CNT][5]  the same five means for a single-material element with
//                       `count` markers (the means are count-dependent in the last bit:
//                       count / (count / s)), so those elements read a cached table row
#define DES_PTAB_CNT 64
struct MatData { const int *markers; const int *mono; const double *props; const double *ptab; const double *pptab; };

__device__ __forceinline__ desk::Mix mix_of(const MatData &md, int nmat, int e)
{
    const int mo = md.mono[e];
    desk::Mix mx;
    if (mo >= 0) { mx.mk = nullptr; mx.mat = mo >> 16; mx.cnt = mo & 0xffff; }
    else         { mx.mk = md.markers + (size_t)e * nmat; mx.mat = -1; mx.cnt = 0; }
    return mx;
}

// the same from a mono[] value the caller has already loaded
__device__ __forceinline__ desk::Mix mix_from_mono(const MatData &md, int nmat, int e, int mo)
{
    desk::Mix mx;
    if (mo >= 0) { mx.mk = nullptr; mx.mat = mo >> 16; mx.cnt = mo & 0xffff; }
    else         { mx.mk = md.markers + (size_t)e * nmat; mx.mat = -1; mx.cnt = 0; }
    return mx;
}

__device__ __forceinline__ ElemProps load_props(const des_params *p, const MatData &md, const desk::Mix &mx, int ne, int e)
{
    ElemProps r;
    if (!md.props) {                                       // nmat == 1: the means are the values (matprops.cxx:118, 136)
        r.bulkm = p->bulk_modulus[0]; r.shearm = p->shear_modulus[0]; r.phi = p->porosity[0];
        r.cp = p->heat_capacity[0]; r.k = p->therm_cond[0];
    } else if (!mx.mk && mx.cnt < DES_PTAB_CNT) {
        const double *t = md.ptab + ((size_t)mx.mat * DES_PTAB_CNT + mx.cnt) * 5;
        r.bulkm = t[0]; r.shearm = t[1]; r.phi = t[2]; r.cp = t[3]; r.k = t[4];
    } else {
        r.bulkm = md.props[e]; r.shearm = md.props[(size_t)ne + e]; r.phi = md.props[(size_t)2*ne + e];
        r.cp = md.props[(size_t)3*ne + e]; r.k = md.props[(size_t)4*ne + e];
    }
    return r;
}

// Young's-modulus "mass" of an element, only read by damping option 4 (geometry.cxx:1832)
__device__ __forceinline__ double elem_ym(const des_params *p, const MatData &md, int ne, int e)
{
    const ElemProps pr = load_props(p, md, mix_of(md, p->nmat, e), ne, e);
    return 9 * pr.bulkm * pr.shearm / (3 * pr.bulkm + pr.shearm) / 4;
}

// refresh_elem_cache (matprops.cxx:259-303): mono[] for every element, props[] for nmat > 1
__global__ void __launch_bounds__(DES_BLOCK)
k_props(const des_params *p, const int *markers, double *props, int *mono, int ne)
{
    int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    const int nmat = p->nmat;
    const int *mk = markers + (size_t)e * nmat;
    int used = 0, mat = 0, cnt = 0;
    for (int m = 0; m < nmat; ++m) if (mk[m] != 0) { ++used; mat = m; cnt = mk[m]; }
    mono[e] = (used == 1 && cnt > 0 && cnt < 65536) ? ((mat << 16) | cnt) : -1;
    if (!props) return;
    props[e]                = desk::harmonic_mean(p->bulk_modulus, mk, nmat);
    props[(size_t)ne + e]   = desk::harmonic_mean(p->shear_modulus, mk, nmat);
    props[(size_t)2*ne + e] = desk::arithmetic_mean(p->porosity, mk, nmat);
    props[(size_t)3*ne + e] = desk::arithmetic_mean(p->heat_capacity, mk, nmat);
    props[(size_t)4*ne + e] = desk::arithmetic_mean(p->therm_cond, mk, nmat);
}

// plastic_props of a single-material element by (material, marker count, weakening regime): desk::plastic_props itself,
// run once per entry with a plastic strain of that regime (des_kernels.hpp)
template <class M>
__global__ void k_pptab(const des_params *p, double *pptab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nmat = p->nmat;
    if (i >= nmat * DES_PPTAB_CNT * 3) return;
    const int regime = i % 3, cnt = (i / 3) % DES_PPTAB_CNT, mat = i / (3 * DES_PPTAB_CNT);
    desk::Mix mx;
    mx.mk = nullptr; mx.mat = mat; mx.cnt = cnt > 0 ? cnt : 1;           // entry 0 is never read
    const double pls = regime == 0 ? p->pls0[mat] - 1.0 : (regime == 1 ? p->pls0[mat] : p->pls1[mat]);
    double *t = pptab + (size_t)i * 5;
    desk::plastic_props<M>(p, mx, pls, t[0], t[1], t[2], t[3], t[4]);
}

// the five means of a single-material element, by (material, marker count)
__global__ void k_ptab(const des_params *p, double *ptab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nmat = p->nmat;
    if (i >= nmat * DES_PTAB_CNT) return;
    const int mat = i / DES_PTAB_CNT, cnt = i % DES_PTAB_CNT;
    int mk[DES_MAX_MAT];
    for (int m = 0; m < DES_MAX_MAT; ++m) mk[m] = 0;
    mk[mat] = cnt > 0 ? cnt : 1;                           // row 0 is never read
    double *t = ptab + (size_t)i * 5;
    t[0] = desk::harmonic_mean(p->bulk_modulus, mk, nmat);
    t[1] = desk::harmonic_mean(p->shear_modulus, mk, nmat);
    t[2] = desk::arithmetic_mean(p->porosity, mk, nmat);
    t[3] = desk::arithmetic_mean(p->heat_capacity, mk, nmat);
    t[4] = desk::arithmetic_mean(p->therm_cond, mk, nmat);
}
