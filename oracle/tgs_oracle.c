/*
 * tgs_oracle.c -- CPU restatement of the reference differentiable Gaussian rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The product (youreditableavatar_amd/csrc) never links it.
 *
 * Parity status: PARITY UNPINNED by the reference's own tests (the reference ships none, SURVEY.md
 * section 4) and the reference itself (CUDA-only) cannot be built in this image.  The restatement is
 * cross-checked against (a) the reference's kernel text executed through a CUDA execution-model
 * emulation (oracle/emu_crosscheck, evidence only, not a reference build) and (b) an independent fp64
 * PyTorch-autograd splat (oracle/torch_splat.py).
 *
 * Reference = /root/reference/Edit_core/thirdparties/diff-gaussian-rasterization  (DGR below),
 * CR = DGR/cuda_rasterizer.  Every function cites the file:line it restates.  Arithmetic is fp32 in
 * the reference's operation order (build with -ffp-contract=off); the only deliberate deviation is
 * that the cross-pixel gradient sums, which the reference forms with fp32 atomicAdd in an undefined
 * order (CR/backward.cu:523-554), are accumulated here in double and rounded once.
 *
 * Matrices follow GLM's column-major convention: m[c][r] is column c, row r
 * (DGR/third_party/glm/glm/detail/type_mat3x3.inl:486-518 for the product order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* `real` is float: the reference's arithmetic.  -DTGS_ORACLE_F64 compiles the SAME text with real = double
 * (libtgs_oracle_f64.so: every array of the interface and of the state is double, the fp32 literals and the fp32
 * depth bits of the sort key stay what they are): the reference's function in exact arithmetic on the same fp32
 * inputs, used by the tests to measure the fp32 reference's own distance from it. */
#ifdef TGS_ORACLE_F64
typedef double real;
#define r_sqrt sqrt
#define r_exp exp
#define r_ceil ceil
#else
typedef float real;
#define r_sqrt sqrtf
#ifdef TGS_ORACLE_EXP2
/* exp the way GPUs evaluate it: 2^(x * log2 e) with the product rounded to fp32 (CUDA's expf: ex2.approx on a scaled argument, documented
 * at up to 2 ulp; glibc's expf is < 1 ulp).  A third legitimate rounding of the reference's arithmetic (libtgs_oracle_ex2.so). */
static inline float exp_via_exp2(float x) { return exp2f(x * 1.44269504088896340736f); }
#define r_exp exp_via_exp2
#else
#define r_exp expf
#endif
#define r_ceil ceilf
#endif
#define RS sizeof(real)

/* The two cut-offs of the compositing loop (forward.cu:340-349, backward.cu:500-504) are DISCONTINUITIES of the reference's function: a pair
 * whose alpha lies within fp32's own evaluation noise of 1/255 is blended or skipped by the last bits of the quadratic form, a pixel whose T
 * lies that close to 1e-4 stops one entry earlier or later -- and at the edge of a large splat one such pair carries dx^2 ~ (3 sigma)^2 of
 * weight into dL_dconic.  Any other legitimate evaluation (the reference's own under nvcc included) decides such pairs the other way.
 * The noise is not a constant: power = -1/2 (A dx^2 + C dy^2) - B dx dy is a sum of terms that cancel, and its fp32 evaluation (and already
 * the fp32 rounding of A, B, C) errs by a few ulp of S = 1/2 (|A| dx^2 + |C| dy^2) + |B dx dy|, i.e. alpha by that much RELATIVELY:
 *   fuzz seed 23 scene 93, pixel (30, 114), Gaussian 2470 (S = 5.5):   alpha * 255 - 1 = -4.5e-7 exact, -8.9e-7 here, +1.8e-7 in the kernels
 *   fuzz seed 37 scene 89, pixel (148, 154), Gaussian 4860 (S = 4458): alpha * 255 - 1 = +8.6e-6 exact, -1.7e-5 here, +7.8e-5 in the kernels
 * -DTGS_ORACLE_CUT=+1 / -1 decides every pair inside the band  |alpha / (1/255) - 1| <= max(1e-6, 8 * 2^-24 * S)  as blended / as skipped
 * (and moves the T cut-off by 1e-6 of its value the same way): libtgs_oracle_in.so / libtgs_oracle_out.so, what the tests take as the
 * reference function's own sensitivity to such decisions (tests/adjudicate.py). */
#define ALPHA_MIN (1.0f / 255.0f)
#ifdef TGS_ORACLE_CUT
#define T_MIN (0.0001f * (1.0f - (TGS_ORACLE_CUT) * 1e-6f))
static inline int alpha_skipped(float alpha, float A, float B, float C, float dx, float dy)
{
    const float S = 0.5f * (fabsf(A) * dx * dx + fabsf(C) * dy * dy) + fabsf(B * dx * dy);
    const float band = fmaxf(1e-6f, 8.0f * 5.9604645e-8f * S);
    return alpha < ALPHA_MIN * (1.0f - (TGS_ORACLE_CUT) * band);
}
#define ALPHA_SKIPPED(alpha, co, dx, dy) alpha_skipped((float)(alpha), (float)(co)[0], (float)(co)[1], (float)(co)[2], (float)(dx), (float)(dy))
#else
#define T_MIN 0.0001f
#define ALPHA_SKIPPED(alpha, co, dx, dy) ((alpha) < ALPHA_MIN)
#endif

/* -DTGS_ORACLE_STATE32 (with -DTGS_ORACLE_F64: libtgs_oracle_f64s.so): exact arithmetic, but the per-Gaussian STATE the reference keeps in
 * fp32 arrays between its kernels -- cov3D, means2D, conic_opacity, rgb (geomState, rasterizer_impl.h:35-49) -- is rounded to fp32 where it
 * is stored, as in the reference.  Its distance from the all-double build is what the fp32 rounding of that state alone does to the result:
 * for splats hundreds of pixels wide, dL_dcov3D / dL_drotations move by 1e-4 with the last bit of cov3D (fuzz seeds 54 / 58, DESIGN.md
 * section 3) -- noise of the reference's own data layout, part of the tests' measure of it. */
#ifdef TGS_ORACLE_STATE32
#define ST32(x) ((real)(float)(x))
#else
#define ST32(x) (x)
#endif

#define BLOCK_X 16 /* CR/config.h:16 */
#define BLOCK_Y 16 /* CR/config.h:17 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)

/* CR/auxiliary.h:22-39 */
static const real SH_C0 = 0.28209479177387814f;
static const real SH_C1 = 0.4886025119029199f;
static const real SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const real SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct { real m[3][3]; } mat3; /* m[col][row] */
typedef struct { real x, y, z; } vec3;

/* glm mat3 * mat3, type_mat3x3.inl:486-518 */
static mat3 m3mul(const mat3* a, const mat3* b)
{
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int w = 0; w < 3; w++)
            r.m[c][w] = a->m[0][w] * b->m[c][0] + a->m[1][w] * b->m[c][1] + a->m[2][w] * b->m[c][2];
    return r;
}
static mat3 m3t(const mat3* a)
{
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int w = 0; w < 3; w++) r.m[c][w] = a->m[w][c];
    return r;
}
/* glm::mat3(a,b,c, d,e,f, g,h,i): arguments fill column 0 first */
static mat3 m3make(real a, real b, real c, real d, real e, real f, real g, real h, real i)
{
    mat3 r = {{{a, b, c}, {d, e, f}, {g, h, i}}};
    return r;
}
static real fminf_(real a, real b) { return a < b ? a : b; }
static real fmaxf_(real a, real b) { return a > b ? a : b; }

/* CR/auxiliary.h:41-44 : literals are double, so the expression is evaluated in double */
static real ndc2Pix(real v, int S) { return (real)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

/* CR/auxiliary.h:46-56 */
static void getRect(real px, real py, int max_radius, uint32_t* rmin, uint32_t* rmax, uint32_t gx,
                    uint32_t gy)
{
    int v;
    v = (int)((px - (real)max_radius) / (real)BLOCK_X); if (v < 0) v = 0;
    rmin[0] = (uint32_t)v < gx ? (uint32_t)v : gx;
    v = (int)((py - (real)max_radius) / (real)BLOCK_Y); if (v < 0) v = 0;
    rmin[1] = (uint32_t)v < gy ? (uint32_t)v : gy;
    v = (int)((px + (real)max_radius + (real)BLOCK_X - 1.0f) / (real)BLOCK_X); if (v < 0) v = 0;
    rmax[0] = (uint32_t)v < gx ? (uint32_t)v : gx;
    v = (int)((py + (real)max_radius + (real)BLOCK_Y - 1.0f) / (real)BLOCK_Y); if (v < 0) v = 0;
    rmax[1] = (uint32_t)v < gy ? (uint32_t)v : gy;
}

/* CR/auxiliary.h:58-77 */
static vec3 transformPoint4x3(vec3 p, const real* m)
{
    vec3 t = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
              m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
    return t;
}
static void transformPoint4x4(vec3 p, const real* m, real* o)
{
    o[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    o[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    o[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    o[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}
/* CR/auxiliary.h:89-97 */
static vec3 transformVec4x3Transpose(vec3 p, const real* m)
{
    vec3 t = {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
              m[8] * p.x + m[9] * p.y + m[10] * p.z};
    return t;
}
/* CR/auxiliary.h:107-117 */
static vec3 dnormvdv3(vec3 v, vec3 dv)
{
    real sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    real invsum32 = 1.0f / r_sqrt(sum2 * sum2 * sum2);
    vec3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

/* ------------------------------------------------------------------------------------------- */
typedef struct tgs_oracle_state {
    int P, D, M, W, H;
    uint32_t gx, gy;
    int64_t R;
    int has_sh, has_colors_precomp, has_cov_precomp;
    real* depths;          /* P   */
    real* means2D;         /* 2P  */
    real* cov3D;           /* 6P  */
    real* conic_opacity;   /* 4P  */
    real* rgb;             /* 3P  */
    uint8_t* clamped;       /* 3P  */
    uint32_t* tiles_touched;/* P   */
    uint32_t* point_offsets;/* P inclusive scan */
    int* radii;             /* P   */
    uint32_t* point_list;   /* R   */
    uint64_t* point_keys;   /* R   */
    uint32_t* ranges;       /* 2T  */
    real* final_T;         /* N   */
    uint32_t* n_contrib;    /* N   */
} tgs_oracle_state;

void tgs_oracle_free(tgs_oracle_state* s)
{
    if (!s) return;
    free(s->depths); free(s->means2D); free(s->cov3D); free(s->conic_opacity); free(s->rgb);
    free(s->clamped); free(s->tiles_touched); free(s->point_offsets); free(s->radii);
    free(s->point_list); free(s->point_keys); free(s->ranges); free(s->final_T); free(s->n_contrib);
    free(s);
}

/* CR/forward.cu:118-152 */
static void computeCov3D(const real* scale, real mod, const real* rot, real* cov3D)
{
    mat3 S = m3make(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * scale[0];
    S.m[1][1] = mod * scale[1];
    S.m[2][2] = mod * scale[2];
    real r = rot[0], x = rot[1], y = rot[2], z = rot[3]; /* not normalised: forward.cu:127 */
    mat3 R = m3make(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                    2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                    2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    mat3 Mx = m3mul(&S, &R);
    mat3 Mt = m3t(&Mx);
    mat3 Sigma = m3mul(&Mt, &Mx);
    cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

/* shared by CR/forward.cu:74-113 and CR/backward.cu:163-198 */
static void cov2d_common(vec3 mean, real fx, real fy, real tan_fovx, real tan_fovy, const real* cov3D,
                         const real* vm, vec3* t_out, real* txtz_o, real* tytz_o, mat3* J, mat3* Wm,
                         mat3* Vrk, mat3* T, mat3* cov)
{
    vec3 t = transformPoint4x3(mean, vm);
    const real limx = 1.3f * tan_fovx;
    const real limy = 1.3f * tan_fovy;
    const real txtz = t.x / t.z;
    const real tytz = t.y / t.z;
    t.x = fminf_(limx, fmaxf_(-limx, txtz)) * t.z;
    t.y = fminf_(limy, fmaxf_(-limy, tytz)) * t.z;
    *J = m3make(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z), 0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z), 0, 0, 0);
    *Wm = m3make(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    *T = m3mul(Wm, J);
    *Vrk = m3make(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    mat3 Tt = m3t(T), Vt = m3t(Vrk);
    mat3 tmp = m3mul(&Tt, &Vt);
    *cov = m3mul(&tmp, T);
    *t_out = t; *txtz_o = txtz; *tytz_o = tytz;
}

/* CR/forward.cu:20-71 */
static void colorFromSH(int idx, int deg, int max_coeffs, const real* means, const real* campos,
                        const real* shs, uint8_t* clamped, real* out)
{
    real dx = means[3 * idx] - campos[0], dy = means[3 * idx + 1] - campos[1], dz = means[3 * idx + 2] - campos[2];
    real len = r_sqrt(dx * dx + dy * dy + dz * dz);
    real x = dx / len, y = dy / len, z = dz / len;
    const real* sh = shs + (size_t)idx * max_coeffs * 3;
    for (int c = 0; c < 3; c++) {
#define SH(k) sh[3 * (k) + c]
        real result = SH_C0 * SH(0);
        if (deg > 0) {
            result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                result = result + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
                         SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
                         SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    result = result + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                             SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                             SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                             SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                             SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        result += 0.5f;
        clamped[3 * idx + c] = (result < 0);
        out[c] = result > 0.0f ? result : 0.0f;
    }
}

static int cmp_key(const void* a, const void* b)
{
    /* elements are {key, seq}: stable order = cub stable radix sort (CR/rasterizer_impl.cu:303-308) */
    const uint64_t* x = (const uint64_t*)a; const uint64_t* y = (const uint64_t*)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    return x[1] < y[1] ? -1 : (x[1] > y[1]);
}

/* CR/rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

/*
 * Forward: CR/rasterizer_impl.cu:198-336 (host pipeline), CR/forward.cu:155-256 (preprocess),
 * CR/rasterizer_impl.cu:70-138 (duplicateWithKeys, identifyTileRanges), CR/forward.cu:261-374 (render).
 * Pointers that the reference receives as nullptr (absent inputs) are NULL here.
 */
tgs_oracle_state* tgs_oracle_forward(int P, int D, int M, const real* background, int W, int H,
                                     const real* means3D, const real* shs, const real* colors_precomp,
                                     const real* opacities, const real* scales, real scale_modifier,
                                     const real* rotations, const real* cov3D_precomp,
                                     const real* viewmatrix, const real* projmatrix, const real* cam_pos,
                                     real tan_fovx, real tan_fovy, real* out_color, int* radii_out)
{
    tgs_oracle_state* s = (tgs_oracle_state*)calloc(1, sizeof(*s));
    const size_t N = (size_t)W * H;
    s->P = P; s->D = D; s->M = M; s->W = W; s->H = H;
    s->gx = (uint32_t)((W + BLOCK_X - 1) / BLOCK_X);
    s->gy = (uint32_t)((H + BLOCK_Y - 1) / BLOCK_Y);
    const size_t T = (size_t)s->gx * s->gy;
    s->has_sh = shs != NULL; s->has_colors_precomp = colors_precomp != NULL; s->has_cov_precomp = cov3D_precomp != NULL;
    size_t Pa = P > 0 ? (size_t)P : 1;
    s->depths = calloc(Pa, RS); s->means2D = calloc(Pa, 2 * RS); s->cov3D = calloc(Pa, 6 * RS);
    s->conic_opacity = calloc(Pa, 4 * RS); s->rgb = calloc(Pa, 3 * RS); s->clamped = calloc(Pa, 3);
    s->tiles_touched = calloc(Pa, 4); s->point_offsets = calloc(Pa, 4); s->radii = calloc(Pa, 4);
    s->ranges = calloc(T ? T : 1, 8); s->final_T = calloc(N ? N : 1, RS); s->n_contrib = calloc(N ? N : 1, 4);

    const real focal_y = H / (2.0f * tan_fovy); /* rasterizer_impl.cu:222-223 */
    const real focal_x = W / (2.0f * tan_fovx);

    /* ---- preprocessCUDA, forward.cu:155-256 ---- */
    for (int idx = 0; idx < P; idx++) {
        s->radii[idx] = 0; s->tiles_touched[idx] = 0;
        vec3 p_orig = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        /* in_frustum, auxiliary.h:139-164 */
        real p_hom[4]; transformPoint4x4(p_orig, projmatrix, p_hom);
        real p_w = 1.0f / (p_hom[3] + 0.0000001f);
        real p_proj_x = p_hom[0] * p_w, p_proj_y = p_hom[1] * p_w;
        vec3 p_view = transformPoint4x3(p_orig, viewmatrix);
        if (p_view.z <= 0.2f) continue;

        const real* cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + 6 * (size_t)idx;
        else {
            computeCov3D(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, s->cov3D + 6 * (size_t)idx);
            for (int k = 0; k < 6; k++) s->cov3D[6 * (size_t)idx + k] = ST32(s->cov3D[6 * (size_t)idx + k]);
            cov3D = s->cov3D + 6 * (size_t)idx;
        }

        vec3 t; real txtz, tytz; mat3 J, Wm, Vrk, Tm, cov;
        cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &t, &txtz, &tytz, &J, &Wm, &Vrk, &Tm, &cov);
        cov.m[0][0] += 0.3f; cov.m[1][1] += 0.3f; /* forward.cu:110-111 */
        real cx = cov.m[0][0], cy = cov.m[0][1], cz = cov.m[1][1];

        real det = (cx * cz - cy * cy);
        if (det == 0.0f) continue;
        real det_inv = 1.f / det;
        real conic[3] = {cz * det_inv, -cy * det_inv, cx * det_inv};

        real mid = 0.5f * (cx + cz);
        real lambda1 = mid + r_sqrt(fmaxf_(0.1f, mid * mid - det));
        real lambda2 = mid - r_sqrt(fmaxf_(0.1f, mid * mid - det));
        real my_radius = r_ceil(3.f * r_sqrt(fmaxf_(lambda1, lambda2)));
        real pix = ndc2Pix(p_proj_x, W), piy = ndc2Pix(p_proj_y, H);
        uint32_t rmin[2], rmax[2];
        getRect(pix, piy, (int)my_radius, rmin, rmax, s->gx, s->gy);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;

        if (!colors_precomp) { colorFromSH(idx, D, M, means3D, cam_pos, shs, s->clamped, s->rgb + 3 * (size_t)idx); for (int k = 0; k < 3; k++) s->rgb[3 * (size_t)idx + k] = ST32(s->rgb[3 * (size_t)idx + k]); }

        s->depths[idx] = p_view.z;
        s->radii[idx] = (int)my_radius;
        s->means2D[2 * idx] = ST32(pix); s->means2D[2 * idx + 1] = ST32(piy);
        s->conic_opacity[4 * idx] = ST32(conic[0]); s->conic_opacity[4 * idx + 1] = ST32(conic[1]);
        s->conic_opacity[4 * idx + 2] = ST32(conic[2]); s->conic_opacity[4 * idx + 3] = opacities[idx];
        s->tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
    }
    if (radii_out) memcpy(radii_out, s->radii, (size_t)P * 4);

    /* ---- InclusiveSum, rasterizer_impl.cu:277-281 ---- */
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += s->tiles_touched[i]; s->point_offsets[i] = acc; }
    s->R = P > 0 ? (int64_t)acc : 0;
    const size_t R = (size_t)s->R;

    /* ---- duplicateWithKeys + stable sort on low 32+bit bits + identifyTileRanges ---- */
    uint64_t* kv = (uint64_t*)malloc((R ? R : 1) * 16);
    for (int idx = 0; idx < P; idx++) {
        if (s->radii[idx] <= 0) continue;
        uint32_t off = idx == 0 ? 0 : s->point_offsets[idx - 1];
        uint32_t rmin[2], rmax[2];
        getRect(s->means2D[2 * idx], s->means2D[2 * idx + 1], s->radii[idx], rmin, rmax, s->gx, s->gy);
        const float d32 = (float)s->depths[idx]; /* the key holds the fp32 bit pattern of the depth */
        uint32_t dbits; memcpy(&dbits, &d32, 4);
        for (uint32_t y = rmin[1]; y < rmax[1]; y++)
            for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
                uint64_t key = (uint64_t)(y * s->gx + x); key <<= 32; key |= dbits;
                kv[2 * (size_t)off] = key; kv[2 * (size_t)off + 1] = ((uint64_t)off << 32) | (uint32_t)idx; off++;
            }
    }
    {
        uint32_t bit = getHigherMsb((uint32_t)T);
        (void)bit; /* keys never have bits above 32+bit set, so the masked sort equals a full-key sort */
    }
    qsort(kv, R, 16, cmp_key);
    s->point_list = (uint32_t*)malloc((R ? R : 1) * 4);
    s->point_keys = (uint64_t*)malloc((R ? R : 1) * 8);
    for (size_t i = 0; i < R; i++) { s->point_keys[i] = kv[2 * i]; s->point_list[i] = (uint32_t)(kv[2 * i + 1] & 0xffffffffu); }
    free(kv);
    for (size_t i = 0; i < R; i++) { /* rasterizer_impl.cu:116-138 */
        uint32_t cur = (uint32_t)(s->point_keys[i] >> 32);
        if (i == 0) s->ranges[2 * cur] = 0;
        else { uint32_t prev = (uint32_t)(s->point_keys[i - 1] >> 32); if (cur != prev) { s->ranges[2 * prev + 1] = (uint32_t)i; s->ranges[2 * cur] = (uint32_t)i; } }
        if (i == R - 1) s->ranges[2 * cur + 1] = (uint32_t)R;
    }

    /* ---- renderCUDA, forward.cu:261-374 ---- */
    const real* features = colors_precomp ? colors_precomp : s->rgb;
    const int gx = (int)s->gx, gy = (int)s->gy;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            uint32_t r0 = s->ranges[2 * (ty * gx + tx)], r1 = s->ranges[2 * (ty * gx + tx) + 1];
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                    if (!(px < W && py < H)) continue;
                    size_t pix_id = (size_t)W * py + px;
                    real pixfx = (real)px, pixfy = (real)py;
                    real Tr = 1.0f; uint32_t contributor = 0, last_contributor = 0; real C[3] = {0, 0, 0};
                    for (uint32_t k = r0; k < r1; k++) {
                        contributor++;
                        uint32_t id = s->point_list[k];
                        real dx = s->means2D[2 * id] - pixfx, dy = s->means2D[2 * id + 1] - pixfy;
                        const real* co = s->conic_opacity + 4 * (size_t)id;
                        real power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        real alpha = fminf_(0.99f, co[3] * r_exp(power));
                        if (ALPHA_SKIPPED(alpha, co, dx, dy)) continue;
                        real test_T = Tr * (1 - alpha);
                        if (test_T < T_MIN) break; /* done = true: later entries never touch this pixel */
                        for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * (size_t)id + ch] * alpha * Tr;
                        Tr = test_T; last_contributor = contributor;
                    }
                    s->final_T[pix_id] = Tr; s->n_contrib[pix_id] = last_contributor;
                    for (int ch = 0; ch < 3; ch++) out_color[ch * N + pix_id] = C[ch] + Tr * background[ch];
                }
        }
    return s;
}

/* CR/backward.cu:20-139 */
static void colorFromSH_bwd(int idx, int deg, int max_coeffs, const real* means, const real* campos,
                            const real* shs, const uint8_t* clamped, const real* dL_dcolor, real* dL_dmeans,
                            real* dL_dshs)
{
    vec3 dir_orig = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    real len = r_sqrt(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    real x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const real* sh = shs + (size_t)idx * max_coeffs * 3;
    real* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
    real dRGB[3];
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? 0.f : 1.f);
    real ddir[3] = {0, 0, 0};
    real gx[3] = {0, 0, 0}, gy[3] = {0, 0, 0}, gz[3] = {0, 0, 0}; /* dRGBdx, dRGBdy, dRGBdz */
#define SH(k) sh[3 * (k) + c]
#define DSH(k, v) for (int c = 0; c < 3; c++) dL_dsh[3 * (k) + c] = (v) * dRGB[c]
    DSH(0, SH_C0);
    if (deg > 0) {
        real d1 = -SH_C1 * y, d2 = SH_C1 * z, d3 = -SH_C1 * x;
        DSH(1, d1); DSH(2, d2); DSH(3, d3);
        for (int c = 0; c < 3; c++) { gx[c] = -SH_C1 * SH(3); gy[c] = -SH_C1 * SH(1); gz[c] = SH_C1 * SH(2); }
        if (deg > 1) {
            real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            real d4 = SH_C2[0] * xy, d5 = SH_C2[1] * yz, d6 = SH_C2[2] * (2.f * zz - xx - yy), d7 = SH_C2[3] * xz, d8 = SH_C2[4] * (xx - yy);
            DSH(4, d4); DSH(5, d5); DSH(6, d6); DSH(7, d7); DSH(8, d8);
            for (int c = 0; c < 3; c++) {
                gx[c] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                gy[c] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                gz[c] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
            }
            if (deg > 2) {
                real d9 = SH_C3[0] * y * (3.f * xx - yy), d10 = SH_C3[1] * xy * z, d11 = SH_C3[2] * y * (4.f * zz - xx - yy);
                real d12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), d13 = SH_C3[4] * x * (4.f * zz - xx - yy);
                real d14 = SH_C3[5] * z * (xx - yy), d15 = SH_C3[6] * x * (xx - 3.f * yy);
                DSH(9, d9); DSH(10, d10); DSH(11, d11); DSH(12, d12); DSH(13, d13); DSH(14, d14); DSH(15, d15);
                for (int c = 0; c < 3; c++) {
                    gx[c] += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy +
                              SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                              SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy));
                    gy[c] += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                              SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
                              SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz + SH_C3[6] * SH(15) * -3.f * 2.f * xy);
                    gz[c] += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                              SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
                              SH_C3[5] * SH(14) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    /* glm::dot = x*x' + y*y' + z*z' left to right */
    ddir[0] = gx[0] * dRGB[0] + gx[1] * dRGB[1] + gx[2] * dRGB[2];
    ddir[1] = gy[0] * dRGB[0] + gy[1] * dRGB[1] + gy[2] * dRGB[2];
    ddir[2] = gz[0] * dRGB[0] + gz[1] * dRGB[1] + gz[2] * dRGB[2];
    vec3 dd = {ddir[0], ddir[1], ddir[2]};
    vec3 dm = dnormvdv3(dir_orig, dd);
    dL_dmeans[3 * idx] += dm.x; dL_dmeans[3 * idx + 1] += dm.y; dL_dmeans[3 * idx + 2] += dm.z;
}

/* CR/backward.cu:278-341 */
static void computeCov3D_bwd(int idx, const real* scale, real mod, const real* rot, const real* dL_dcov3Ds,
                             real* dL_dscales, real* dL_drots)
{
    real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = m3make(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                    2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                    2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    mat3 S = m3make(1, 0, 0, 0, 1, 0, 0, 0, 1);
    real sx = mod * scale[0], sy = mod * scale[1], sz = mod * scale[2];
    S.m[0][0] = sx; S.m[1][1] = sy; S.m[2][2] = sz;
    mat3 Mx = m3mul(&S, &R);
    const real* d = dL_dcov3Ds + 6 * (size_t)idx;
    mat3 dSig = m3make(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]);
    /* dL_dM = 2.0f * M * dL_dSigma : (2.0f * M) first (scalar*mat), then the product */
    mat3 M2; for (int c = 0; c < 3; c++) for (int w = 0; w < 3; w++) M2.m[c][w] = 2.0f * Mx.m[c][w];
    mat3 dM = m3mul(&M2, &dSig);
    mat3 Rt = m3t(&R), dMt = m3t(&dM);
    real* ds = dL_dscales + 3 * (size_t)idx;
    ds[0] = Rt.m[0][0] * dMt.m[0][0] + Rt.m[0][1] * dMt.m[0][1] + Rt.m[0][2] * dMt.m[0][2];
    ds[1] = Rt.m[1][0] * dMt.m[1][0] + Rt.m[1][1] * dMt.m[1][1] + Rt.m[1][2] * dMt.m[1][2];
    ds[2] = Rt.m[2][0] * dMt.m[2][0] + Rt.m[2][1] * dMt.m[2][1] + Rt.m[2][2] * dMt.m[2][2];
    for (int w = 0; w < 3; w++) { dMt.m[0][w] *= sx; dMt.m[1][w] *= sy; dMt.m[2][w] *= sz; }
    real* dq = dL_drots + 4 * (size_t)idx;
#define A(c, w) dMt.m[c][w]
    dq[0] = 2 * z * (A(0, 1) - A(1, 0)) + 2 * y * (A(2, 0) - A(0, 2)) + 2 * x * (A(1, 2) - A(2, 1));
    dq[1] = 2 * y * (A(1, 0) + A(0, 1)) + 2 * z * (A(2, 0) + A(0, 2)) + 2 * r * (A(1, 2) - A(2, 1)) - 4 * x * (A(2, 2) + A(1, 1));
    dq[2] = 2 * x * (A(1, 0) + A(0, 1)) + 2 * r * (A(2, 0) - A(0, 2)) + 2 * z * (A(1, 2) + A(2, 1)) - 4 * y * (A(2, 2) + A(0, 0));
    dq[3] = 2 * r * (A(0, 1) - A(1, 0)) + 2 * x * (A(2, 0) + A(0, 2)) + 2 * y * (A(1, 2) + A(2, 1)) - 4 * z * (A(1, 1) + A(0, 0));
#undef A
}

/*
 * Backward: CR/rasterizer_impl.cu:340-434; render CR/backward.cu:399-557; computeCov2DCUDA :144-274;
 * preprocessCUDA :346-396.  All outputs must be zero-initialised by the caller exactly as
 * DGR/rasterize_points.cu:151-159 does (torch::zeros); dL_dconic is [P,4], dL_dmean2D is [P,3].
 */
void tgs_oracle_backward(const tgs_oracle_state* s, const real* background, const real* means3D,
                         const real* shs, const real* colors_precomp, const real* scales, real scale_modifier,
                         const real* rotations, const real* cov3D_precomp, const real* viewmatrix,
                         const real* projmatrix, const real* campos, real tan_fovx, real tan_fovy,
                         const real* dL_dpix, real* dL_dmean2D, real* dL_dconic, real* dL_dopacity,
                         real* dL_dcolor, real* dL_dmean3D, real* dL_dcov3D, real* dL_dsh, real* dL_dscale,
                         real* dL_drot)
{
    const int P = s->P, W = s->W, H = s->H, D = s->D, M = s->M;
    const size_t N = (size_t)W * H;
    const real focal_y = H / (2.0f * tan_fovy);
    const real focal_x = W / (2.0f * tan_fovx);
    const real* colors = colors_precomp ? colors_precomp : s->rgb;
    const int gx = (int)s->gx, gy = (int)s->gy;

    /* 9 accumulators per Gaussian, double (see header): col3, mean2D xy, conic xyw, opacity */
    double* acc = (double*)calloc((size_t)(P > 0 ? P : 1) * 9, sizeof(double));
#ifdef TGS_ORACLE_F32_ACCUM
    /* The reference's own accumulation: fp32 atomicAdd, order undefined (CR/backward.cu:523-554).  Here in a FIXED order so that tests are
     * reproducible: per (tile, list entry) an fp32 sum over the tile's pixels in pixel order, then per Gaussian an fp32 sum over its
     * instances in list order.  One more legitimate rounding of the reference's arithmetic (libtgs_oracle_fma.so). */
    real* part = (real*)calloc((size_t)(s->R > 0 ? s->R : 1) * 9, sizeof(real));
#endif
    const real ddelx_dx = (real)(0.5 * W), ddely_dy = (real)(0.5 * H);

#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            uint32_t r0 = s->ranges[2 * (ty * gx + tx)], r1 = s->ranges[2 * (ty * gx + tx) + 1];
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                    if (!(px < W && py < H)) continue;
                    size_t pix_id = (size_t)W * py + px;
                    real pixfx = (real)px, pixfy = (real)py;
                    const real T_final = s->final_T[pix_id];
                    real Tr = T_final;
                    uint32_t contributor = r1 - r0;
                    const uint32_t last_contributor = s->n_contrib[pix_id];
                    real accum_rec[3] = {0, 0, 0}, dL_dpixel[3], last_alpha = 0, last_color[3] = {0, 0, 0};
                    for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpix[i * N + pix_id];
                    for (uint32_t k = r1; k-- > r0;) { /* back to front: backward.cu:472-488 */
                        contributor--;
                        if (contributor >= last_contributor) continue;
                        uint32_t id = s->point_list[k];
                        real dx = s->means2D[2 * id] - pixfx, dy = s->means2D[2 * id + 1] - pixfy;
                        const real* co = s->conic_opacity + 4 * (size_t)id;
                        real power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const real G = r_exp(power);
                        const real alpha = fminf_(0.99f, co[3] * G);
                        if (ALPHA_SKIPPED(alpha, co, dx, dy)) continue;
                        Tr = Tr / (1.f - alpha);
                        const real dchannel_dcolor = alpha * Tr;
                        real dL_dalpha = 0.0f;
#ifdef TGS_ORACLE_F32_ACCUM
                        real* a = part + 9 * (size_t)k;
#define ACC_ADD(j, v) a[j] += (v)
#else
                        double* a = acc + 9 * (size_t)id;
#define ACC_ADD(j, v) _Pragma("omp atomic") a[j] += (double)(v)
#endif
                        for (int ch = 0; ch < 3; ch++) {
                            const real c = colors[3 * (size_t)id + ch];
                            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                            last_color[ch] = c;
                            const real dL_dchannel = dL_dpixel[ch];
                            dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                            real v = dchannel_dcolor * dL_dchannel;
                            ACC_ADD(ch, v);
                        }
                        dL_dalpha *= Tr;
                        last_alpha = alpha;
                        real bg_dot_dpixel = 0;
                        for (int i = 0; i < 3; i++) bg_dot_dpixel += background[i] * dL_dpixel[i];
                        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                        const real dL_dG = co[3] * dL_dalpha;
                        const real gdx = G * dx, gdy = G * dy;
                        const real dG_ddelx = -gdx * co[0] - gdy * co[1];
                        const real dG_ddely = -gdy * co[2] - gdx * co[1];
                        real v3 = dL_dG * dG_ddelx * ddelx_dx, v4 = dL_dG * dG_ddely * ddely_dy;
                        real v5 = -0.5f * gdx * dx * dL_dG, v6 = -0.5f * gdx * dy * dL_dG, v7 = -0.5f * gdy * dy * dL_dG;
                        real v8 = G * dL_dalpha;
                        ACC_ADD(3, v3);
                        ACC_ADD(4, v4);
                        ACC_ADD(5, v5);
                        ACC_ADD(6, v6);
                        ACC_ADD(7, v7);
                        ACC_ADD(8, v8);
                    }
                }
        }
#undef ACC_ADD
#ifdef TGS_ORACLE_F32_ACCUM
    {
        real* acc32 = (real*)calloc((size_t)(P > 0 ? P : 1) * 9, sizeof(real));
        for (int64_t k = 0; k < s->R; k++) {
            const uint32_t id = s->point_list[k];
            for (int j = 0; j < 9; j++) acc32[9 * (size_t)id + j] += part[9 * (size_t)k + j];
        }
        for (size_t i = 0; i < (size_t)P * 9; i++) acc[i] = (double)acc32[i];
        free(acc32); free(part);
    }
#endif
    for (int i = 0; i < P; i++) {
        const double* a = acc + 9 * (size_t)i;
        dL_dcolor[3 * i] += (real)a[0]; dL_dcolor[3 * i + 1] += (real)a[1]; dL_dcolor[3 * i + 2] += (real)a[2];
        dL_dmean2D[3 * i] += (real)a[3]; dL_dmean2D[3 * i + 1] += (real)a[4];
        dL_dconic[4 * i] += (real)a[5]; dL_dconic[4 * i + 1] += (real)a[6]; dL_dconic[4 * i + 3] += (real)a[7];
        dL_dopacity[i] += (real)a[8];
    }
    free(acc);

    const real* cov3Ds = cov3D_precomp ? cov3D_precomp : s->cov3D;
    /* ---- computeCov2DCUDA, backward.cu:144-274 ---- */
    for (int idx = 0; idx < P; idx++) {
        if (!(s->radii[idx] > 0)) continue;
        const real* cov3D = cov3Ds + 6 * (size_t)idx;
        vec3 mean = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        real dLc[3] = {dL_dconic[4 * idx], dL_dconic[4 * idx + 1], dL_dconic[4 * idx + 3]};
        vec3 t; real txtz, tytz; mat3 J, Wm, Vrk, T, cov2D;
        cov2d_common(mean, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &t, &txtz, &tytz, &J, &Wm, &Vrk, &T, &cov2D);
        const real limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
        const real x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const real y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        real a = cov2D.m[0][0] += 0.3f;
        real b = cov2D.m[0][1];
        real c = cov2D.m[1][1] += 0.3f;
        real denom = a * c - b * b;
        real dL_da = 0, dL_db = 0, dL_dc = 0;
        real denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        real* dcov = dL_dcov3D + 6 * (size_t)idx;
#define Tm(c_, r_) T.m[c_][r_]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dLc[0] + 2 * b * c * dLc[1] + (denom - a * c) * dLc[2]);
            dL_dc = denom2inv * (-a * a * dLc[2] + 2 * a * b * dLc[1] + (denom - a * c) * dLc[0]);
            dL_db = denom2inv * 2 * (b * c * dLc[0] - (denom + 2 * b * b) * dLc[1] + a * b * dLc[2]);
            dcov[0] = (Tm(0, 0) * Tm(0, 0) * dL_da + Tm(0, 0) * Tm(1, 0) * dL_db + Tm(1, 0) * Tm(1, 0) * dL_dc);
            dcov[3] = (Tm(0, 1) * Tm(0, 1) * dL_da + Tm(0, 1) * Tm(1, 1) * dL_db + Tm(1, 1) * Tm(1, 1) * dL_dc);
            dcov[5] = (Tm(0, 2) * Tm(0, 2) * dL_da + Tm(0, 2) * Tm(1, 2) * dL_db + Tm(1, 2) * Tm(1, 2) * dL_dc);
            dcov[1] = 2 * Tm(0, 0) * Tm(0, 1) * dL_da + (Tm(0, 0) * Tm(1, 1) + Tm(0, 1) * Tm(1, 0)) * dL_db + 2 * Tm(1, 0) * Tm(1, 1) * dL_dc;
            dcov[2] = 2 * Tm(0, 0) * Tm(0, 2) * dL_da + (Tm(0, 0) * Tm(1, 2) + Tm(0, 2) * Tm(1, 0)) * dL_db + 2 * Tm(1, 0) * Tm(1, 2) * dL_dc;
            dcov[4] = 2 * Tm(0, 2) * Tm(0, 1) * dL_da + (Tm(0, 1) * Tm(1, 2) + Tm(0, 2) * Tm(1, 1)) * dL_db + 2 * Tm(1, 1) * Tm(1, 2) * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dcov[i] = 0;
        }
#define V(c_, r_) Vrk.m[c_][r_]
        real dL_dT00 = 2 * (Tm(0, 0) * V(0, 0) + Tm(0, 1) * V(0, 1) + Tm(0, 2) * V(0, 2)) * dL_da + (Tm(1, 0) * V(0, 0) + Tm(1, 1) * V(0, 1) + Tm(1, 2) * V(0, 2)) * dL_db;
        real dL_dT01 = 2 * (Tm(0, 0) * V(1, 0) + Tm(0, 1) * V(1, 1) + Tm(0, 2) * V(1, 2)) * dL_da + (Tm(1, 0) * V(1, 0) + Tm(1, 1) * V(1, 1) + Tm(1, 2) * V(1, 2)) * dL_db;
        real dL_dT02 = 2 * (Tm(0, 0) * V(2, 0) + Tm(0, 1) * V(2, 1) + Tm(0, 2) * V(2, 2)) * dL_da + (Tm(1, 0) * V(2, 0) + Tm(1, 1) * V(2, 1) + Tm(1, 2) * V(2, 2)) * dL_db;
        real dL_dT10 = 2 * (Tm(1, 0) * V(0, 0) + Tm(1, 1) * V(0, 1) + Tm(1, 2) * V(0, 2)) * dL_dc + (Tm(0, 0) * V(0, 0) + Tm(0, 1) * V(0, 1) + Tm(0, 2) * V(0, 2)) * dL_db;
        real dL_dT11 = 2 * (Tm(1, 0) * V(1, 0) + Tm(1, 1) * V(1, 1) + Tm(1, 2) * V(1, 2)) * dL_dc + (Tm(0, 0) * V(1, 0) + Tm(0, 1) * V(1, 1) + Tm(0, 2) * V(1, 2)) * dL_db;
        real dL_dT12 = 2 * (Tm(1, 0) * V(2, 0) + Tm(1, 1) * V(2, 1) + Tm(1, 2) * V(2, 2)) * dL_dc + (Tm(0, 0) * V(2, 0) + Tm(0, 1) * V(2, 1) + Tm(0, 2) * V(2, 2)) * dL_db;
#undef V
#undef Tm
#define Wg(c_, r_) Wm.m[c_][r_]
        real dL_dJ00 = Wg(0, 0) * dL_dT00 + Wg(0, 1) * dL_dT01 + Wg(0, 2) * dL_dT02;
        real dL_dJ02 = Wg(2, 0) * dL_dT00 + Wg(2, 1) * dL_dT01 + Wg(2, 2) * dL_dT02;
        real dL_dJ11 = Wg(1, 0) * dL_dT10 + Wg(1, 1) * dL_dT11 + Wg(1, 2) * dL_dT12;
        real dL_dJ12 = Wg(2, 0) * dL_dT10 + Wg(2, 1) * dL_dT11 + Wg(2, 2) * dL_dT12;
#undef Wg
        real tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        real dL_dtx = x_grad_mul * -focal_x * tz2 * dL_dJ02;
        real dL_dty = y_grad_mul * -focal_y * tz2 * dL_dJ12;
        real dL_dtz = -focal_x * tz2 * dL_dJ00 - focal_y * tz2 * dL_dJ11 + (2 * focal_x * t.x) * tz3 * dL_dJ02 + (2 * focal_y * t.y) * tz3 * dL_dJ12;
        vec3 dt = {dL_dtx, dL_dty, dL_dtz};
        vec3 dm = transformVec4x3Transpose(dt, viewmatrix);
        dL_dmean3D[3 * idx] = dm.x; dL_dmean3D[3 * idx + 1] = dm.y; dL_dmean3D[3 * idx + 2] = dm.z; /* assignment, :273 */
    }

    /* ---- preprocessCUDA backward, backward.cu:346-396 ---- */
    for (int idx = 0; idx < P; idx++) {
        if (!(s->radii[idx] > 0)) continue;
        vec3 m = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        const real* proj = projmatrix;
        real m_hom[4]; transformPoint4x4(m, proj, m_hom);
        real m_w = 1.0f / (m_hom[3] + 0.0000001f);
        real mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
        real mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
        real g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
        real dmx = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        real dmy = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        real dmz = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        dL_dmean3D[3 * idx] += dmx; dL_dmean3D[3 * idx + 1] += dmy; dL_dmean3D[3 * idx + 2] += dmz;
        if (shs) colorFromSH_bwd(idx, D, M, means3D, campos, shs, s->clamped, dL_dcolor, dL_dmean3D, dL_dsh);
        if (scales) computeCov3D_bwd(idx, scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, dL_dcov3D, dL_dscale, dL_drot);
    }
}

#ifndef TGS_ORACLE_F64
/*
 * The per-Gaussian half of the backward (computeCov2DCUDA + preprocessCUDA, backward.cu:144-396) evaluated in
 * DOUBLE on the same fp32 inputs (means, stored fp32 cov3D, the fp32 dL_dconic / dL_dmean2D / dL_dcolor of the
 * render pass).  Not a restatement of the reference's arithmetic: it measures how much of the fp32 result of those
 * formulas is rounding noise (the (denom - a*c) and T*T cancellations), which tests use to scale the tolerance of
 * dL_dmeans3D / dL_dcov3D / dL_dscales / dL_drotations where no fixture carries that information.
 */
typedef struct { double m[3][3]; } dmat3;
static dmat3 dm3mul(const dmat3* a, const dmat3* b)
{
    dmat3 r;
    for (int c = 0; c < 3; c++) for (int w = 0; w < 3; w++) r.m[c][w] = a->m[0][w] * b->m[c][0] + a->m[1][w] * b->m[c][1] + a->m[2][w] * b->m[c][2];
    return r;
}
static dmat3 dm3t(const dmat3* a) { dmat3 r; for (int c = 0; c < 3; c++) for (int w = 0; w < 3; w++) r.m[c][w] = a->m[w][c]; return r; }

void tgs_oracle_backward_pergauss_f64(const tgs_oracle_state* s, const float* means3D, const float* shs, const float* scales,
                                      float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                      const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                                      const float* dL_dmean2D, const float* dL_dconic, const float* dL_dcolor,
                                      float* dL_dmean3D, float* dL_dcov3D, float* dL_dscale, float* dL_drot)
{
    const int P = s->P, W = s->W, H = s->H, D = s->D, M = s->M;
    const double fy = (double)(H / (2.0f * tan_fovy)), fx = (double)(W / (2.0f * tan_fovx));
    const float* cov3Ds = cov3D_precomp ? cov3D_precomp : s->cov3D;
    const float* vm = viewmatrix; const float* pj = projmatrix;
    for (int idx = 0; idx < P; idx++) {
        double dmean[3] = {0, 0, 0}, dcov[6] = {0, 0, 0, 0, 0, 0};
        if (s->radii[idx] > 0) {
            const double mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
            const float* c3 = cov3Ds + 6 * (size_t)idx;
            double tx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12], ty = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
            const double tz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
            const double limx = (double)(1.3f * tan_fovx), limy = (double)(1.3f * tan_fovy), txtz = tx / tz, tytz = ty / tz;
            tx = (txtz < -limx ? -limx : txtz > limx ? limx : txtz) * tz;
            ty = (tytz < -limy ? -limy : tytz > limy ? limy : tytz) * tz;
            const double xg = (txtz < -limx || txtz > limx) ? 0 : 1, yg = (tytz < -limy || tytz > limy) ? 0 : 1;
            dmat3 J = {{{fx / tz, 0, -(fx * tx) / (tz * tz)}, {0, fy / tz, -(fy * ty) / (tz * tz)}, {0, 0, 0}}};
            dmat3 Wm = {{{vm[0], vm[4], vm[8]}, {vm[1], vm[5], vm[9]}, {vm[2], vm[6], vm[10]}}};
            dmat3 V = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
            dmat3 T = dm3mul(&Wm, &J), Tt = dm3t(&T), Vt = dm3t(&V), tmp = dm3mul(&Tt, &Vt), c2 = dm3mul(&tmp, &T);
            const double a = c2.m[0][0] + (double)0.3f, b = c2.m[0][1], c = c2.m[1][1] + (double)0.3f;
            const double g0 = dL_dconic[4 * idx], g1 = dL_dconic[4 * idx + 1], g2 = dL_dconic[4 * idx + 3];
            const double denom = a * c - b * b, d2i = 1.0 / (denom * denom + (double)0.0000001f);
            double dLa = 0, dLb = 0, dLc = 0;
#define TT(c_, r_) T.m[c_][r_]
#define VV(c_, r_) V.m[c_][r_]
            if (d2i != 0) {
                dLa = d2i * (-c * c * g0 + 2 * b * c * g1 + (denom - a * c) * g2);
                dLc = d2i * (-a * a * g2 + 2 * a * b * g1 + (denom - a * c) * g0);
                dLb = d2i * 2 * (b * c * g0 - (denom + 2 * b * b) * g1 + a * b * g2);
                dcov[0] = TT(0, 0) * TT(0, 0) * dLa + TT(0, 0) * TT(1, 0) * dLb + TT(1, 0) * TT(1, 0) * dLc;
                dcov[3] = TT(0, 1) * TT(0, 1) * dLa + TT(0, 1) * TT(1, 1) * dLb + TT(1, 1) * TT(1, 1) * dLc;
                dcov[5] = TT(0, 2) * TT(0, 2) * dLa + TT(0, 2) * TT(1, 2) * dLb + TT(1, 2) * TT(1, 2) * dLc;
                dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dLa + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dLb + 2 * TT(1, 0) * TT(1, 1) * dLc;
                dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dLa + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dLb + 2 * TT(1, 0) * TT(1, 2) * dLc;
                dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dLa + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dLb + 2 * TT(1, 1) * TT(1, 2) * dLc;
            }
            double dT[2][3];
            for (int k = 0; k < 3; k++) {
                const double r0 = TT(0, 0) * VV(k, 0) + TT(0, 1) * VV(k, 1) + TT(0, 2) * VV(k, 2), r1 = TT(1, 0) * VV(k, 0) + TT(1, 1) * VV(k, 1) + TT(1, 2) * VV(k, 2);
                dT[0][k] = 2 * r0 * dLa + r1 * dLb;
                dT[1][k] = 2 * r1 * dLc + r0 * dLb;
            }
#undef TT
#undef VV
            const double dJ00 = Wm.m[0][0] * dT[0][0] + Wm.m[0][1] * dT[0][1] + Wm.m[0][2] * dT[0][2];
            const double dJ02 = Wm.m[2][0] * dT[0][0] + Wm.m[2][1] * dT[0][1] + Wm.m[2][2] * dT[0][2];
            const double dJ11 = Wm.m[1][0] * dT[1][0] + Wm.m[1][1] * dT[1][1] + Wm.m[1][2] * dT[1][2];
            const double dJ12 = Wm.m[2][0] * dT[1][0] + Wm.m[2][1] * dT[1][1] + Wm.m[2][2] * dT[1][2];
            const double z1 = 1.0 / tz, z2 = z1 * z1, z3 = z2 * z1;
            const double dtx = xg * -fx * z2 * dJ02, dty = yg * -fy * z2 * dJ12;
            const double dtz = -fx * z2 * dJ00 - fy * z2 * dJ11 + (2 * fx * tx) * z3 * dJ02 + (2 * fy * ty) * z3 * dJ12;
            dmean[0] = vm[0] * dtx + vm[1] * dty + vm[2] * dtz; dmean[1] = vm[4] * dtx + vm[5] * dty + vm[6] * dtz; dmean[2] = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;
            /* projection part, backward.cu:369-387 */
            const double mw = 1.0 / ((pj[3] * mx + pj[7] * my + pj[11] * mz + pj[15]) + (double)0.0000001f);
            const double mul1 = (pj[0] * mx + pj[4] * my + pj[8] * mz + pj[12]) * mw * mw, mul2 = (pj[1] * mx + pj[5] * my + pj[9] * mz + pj[13]) * mw * mw;
            const double g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
            dmean[0] += (pj[0] * mw - pj[3] * mul1) * g2x + (pj[1] * mw - pj[3] * mul2) * g2y;
            dmean[1] += (pj[4] * mw - pj[7] * mul1) * g2x + (pj[5] * mw - pj[7] * mul2) * g2y;
            dmean[2] += (pj[8] * mw - pj[11] * mul1) * g2x + (pj[9] * mw - pj[11] * mul2) * g2y;
            if (shs) {   /* the view-direction term of backward.cu:130-138; reuses the fp32 routine's structure in double via finite differences-free algebra */
                float tmp_dm[3] = {0, 0, 0};
                float* dsh_tmp = (float*)calloc((size_t)M * 3 + 1, 4);
                /* colorFromSH_bwd writes at index idx: give it arrays offset so that idx lands on our temporaries */
                colorFromSH_bwd(idx, D, M, means3D, campos, shs, s->clamped, dL_dcolor, tmp_dm - 3 * (size_t)idx, dsh_tmp - (size_t)idx * M * 3);
                dmean[0] += tmp_dm[0]; dmean[1] += tmp_dm[1]; dmean[2] += tmp_dm[2];
                free(dsh_tmp);
            }
            if (scales) {
                const double r = rotations[4 * idx], x = rotations[4 * idx + 1], y = rotations[4 * idx + 2], z = rotations[4 * idx + 3];
                dmat3 R = {{{1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)}, {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
                            {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}}};
                const double sx = (double)scale_modifier * scales[3 * idx], sy = (double)scale_modifier * scales[3 * idx + 1], sz = (double)scale_modifier * scales[3 * idx + 2];
                dmat3 S = {{{sx, 0, 0}, {0, sy, 0}, {0, 0, sz}}};
                dmat3 Mx = dm3mul(&S, &R);
                dmat3 dSig = {{{dcov[0], 0.5 * dcov[1], 0.5 * dcov[2]}, {0.5 * dcov[1], dcov[3], 0.5 * dcov[4]}, {0.5 * dcov[2], 0.5 * dcov[4], dcov[5]}}};
                dmat3 M2; for (int c_ = 0; c_ < 3; c_++) for (int w = 0; w < 3; w++) M2.m[c_][w] = 2.0 * Mx.m[c_][w];
                dmat3 dM = dm3mul(&M2, &dSig), Rt = dm3t(&R), dMt = dm3t(&dM);
                for (int k = 0; k < 3; k++) dL_dscale[3 * idx + k] = (float)(Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2]);
                for (int w = 0; w < 3; w++) { dMt.m[0][w] *= sx; dMt.m[1][w] *= sy; dMt.m[2][w] *= sz; }
#define A(c_, w_) dMt.m[c_][w_]
                dL_drot[4 * idx] = (float)(2 * z * (A(0, 1) - A(1, 0)) + 2 * y * (A(2, 0) - A(0, 2)) + 2 * x * (A(1, 2) - A(2, 1)));
                dL_drot[4 * idx + 1] = (float)(2 * y * (A(1, 0) + A(0, 1)) + 2 * z * (A(2, 0) + A(0, 2)) + 2 * r * (A(1, 2) - A(2, 1)) - 4 * x * (A(2, 2) + A(1, 1)));
                dL_drot[4 * idx + 2] = (float)(2 * x * (A(1, 0) + A(0, 1)) + 2 * r * (A(2, 0) - A(0, 2)) + 2 * z * (A(1, 2) + A(2, 1)) - 4 * y * (A(2, 2) + A(0, 0)));
                dL_drot[4 * idx + 3] = (float)(2 * r * (A(0, 1) - A(1, 0)) + 2 * x * (A(2, 0) + A(0, 2)) + 2 * y * (A(1, 2) + A(2, 1)) - 4 * z * (A(1, 1) + A(0, 0)));
#undef A
            }
        }
        for (int k = 0; k < 3; k++) dL_dmean3D[3 * idx + k] = (float)dmean[k];
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * idx + k] = (float)dcov[k];
    }
}

#endif /* !TGS_ORACLE_F64 */

/* CR/rasterizer_impl.cu:54-66,141-153 */
void tgs_oracle_mark_visible(int P, const real* means3D, const real* viewmatrix, const real* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    for (int idx = 0; idx < P; idx++) {
        vec3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        vec3 pv = transformPoint4x3(p, viewmatrix);
        present[idx] = !(pv.z <= 0.2f);
    }
}

int64_t tgs_oracle_num_rendered(const tgs_oracle_state* s) { return s->R; }
int tgs_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* field access for tests: returns pointer and element count */
const void* tgs_oracle_field(const tgs_oracle_state* s, const char* name, int64_t* count)
{
    const size_t P = (size_t)s->P, N = (size_t)s->W * s->H, T = (size_t)s->gx * s->gy;
#define F(n, p, c) if (!strcmp(name, n)) { *count = (int64_t)(c); return (const void*)(p); }
    F("depths", s->depths, P) F("means2D", s->means2D, 2 * P) F("cov3D", s->cov3D, 6 * P)
    F("conic_opacity", s->conic_opacity, 4 * P) F("rgb", s->rgb, 3 * P) F("clamped", s->clamped, 3 * P)
    F("tiles_touched", s->tiles_touched, P) F("point_offsets", s->point_offsets, P) F("radii", s->radii, P)
    F("point_list", s->point_list, s->R) F("point_keys", s->point_keys, s->R) F("ranges", s->ranges, 2 * T)
    F("final_T", s->final_T, N) F("n_contrib", s->n_contrib, N)
#undef F
    *count = -1;
    return NULL;
}
