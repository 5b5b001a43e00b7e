// vio_device_math.h — small fp64 helpers used by the gfx950 kernels (device only).
// Formulas follow the reference's Eigen/Sophus expressions; each cites where it is used there.
#ifndef VIO_DEVICE_MATH_H
#define VIO_DEVICE_MATH_H

#include <hip/hip_runtime.h>

#define DEV __device__ __forceinline__

// Eigen::QuaternionBase::toRotationMatrix on (x,y,z,w), row-major
DEV void d_quat_to_R(const double *q, double *R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

DEV void d_m3_mul(const double *A, const double *B, double *C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// C = A^T * B
DEV void d_m3_tmul(const double *A, const double *B, double *C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
DEV void d_m3_vec(const double *A, const double *v, double *o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
DEV void d_m3_tvec(const double *A, const double *v, double *o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}

struct dquat { double x, y, z, w; };
DEV dquat d_qmul(dquat a, dquat b) {       // Eigen quaternion product
    dquat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
DEV dquat d_qinv(dquat q) {                // Eigen::QuaternionBase::inverse
    const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    dquat r = {0, 0, 0, 0};
    if (n2 > 0) { r.x = -q.x / n2; r.y = -q.y / n2; r.z = -q.z / n2; r.w = q.w / n2; }
    return r;
}
DEV void d_qrot(dquat q, const double *v, double *o) {   // Eigen _transformVector
    double ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
    ux += ux; uy += uy; uz += uz;
    o[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
    o[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
    o[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}
DEV dquat d_qload(const double *p) { dquat q = {p[3], p[4], p[5], p[6]}; return q; }

// VertexPose::Plus (vertex_pose.cc:7-19) with Sophus::SO3::exp (so3.hpp:393-419, ctor normalise :683-685)
DEV void d_pose_plus(const double *p, const double *d, double *o) {
    o[0] = p[0] + d[0]; o[1] = p[1] + d[1]; o[2] = p[2] + d[2];
    const double ox = d[3], oy = d[4], oz = d[5];
    const double theta_sq = ox * ox + oy * oy + oz * oz;
    const double theta = sqrt(theta_sq);
    const double half_theta = 0.5 * theta;
    double imag, real;
    if (theta < 1e-10) {
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        imag = sin(half_theta) / theta;
        real = cos(half_theta);
    }
    dquat e = {imag * ox, imag * oy, imag * oz, real};
    const double n = sqrt(e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w);
    e.x /= n; e.y /= n; e.z /= n; e.w /= n;
    dquat q = {p[3], p[4], p[5], p[6]};
    q = d_qmul(q, e);      // the reference discards q.normalized() (vertex_pose.cc:12)
    o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
}

// LossFunction::Compute (loss_function.cc:9-47); type 0 = no loss object
DEV void d_loss(int type, double delta, double e2, double &r0, double &r1, double &r2) {
    if (type == 2) {            // Cauchy
        const double dsqr = delta * delta, rec = 1. / dsqr, aux = rec * e2 + 1.0;
        r0 = dsqr * log(aux); r1 = 1. / aux; r2 = -rec * (r1 * r1);
    } else if (type == 1) {     // Huber
        const double dsqr = delta * delta;
        if (e2 <= dsqr) { r0 = e2; r1 = 1.; r2 = 0.; }
        else { const double s = sqrt(e2); r0 = 2 * s * delta - dsqr; r1 = delta / s; r2 = -0.5 * r1 / e2; }
    } else if (type == 3) {     // Tukey
        const double e = sqrt(e2), d2 = delta * delta;
        if (e <= delta) { const double aux = e2 / d2, u = 1. - aux; r0 = d2 * (1. - u * u * u) / 3.; r1 = u * u; r2 = -2. * u / d2; }
        else { r0 = d2 / 3.; r1 = 0; r2 = 0; }
    } else { r0 = e2; r1 = 1; r2 = 0; }
}

// Wave-wide (64 lanes) sum of a double with DPP moves: quad swaps, half-row and row mirrors, then the two row
// broadcasts gfx9 has (row_bcast:15 / row_bcast:31).  Six steps of (2 x v_mov_dpp + v_add_f64) instead of the
// ds_bpermute round trips __shfl_xor costs for 64-bit values.  The total ends up in lane 63; fixed order.
template <int CTRL, int ROW_MASK = 0xf>
DEV double d_dpp_mov(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
DEV double d_wave_sum_to_lane63(double v) {
    v += d_dpp_mov<0xB1>(v);            // quad_perm:[1,0,3,2]
    v += d_dpp_mov<0x4E>(v);            // quad_perm:[2,3,0,1]
    v += d_dpp_mov<0x141>(v);           // row_half_mirror
    v += d_dpp_mov<0x140>(v);           // row_mirror: every lane of a row of 16 now holds the row's sum
    v += d_dpp_mov<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v += d_dpp_mov<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;                           // lane 63 holds the wave's sum
}

DEV double d_wave_max_to_lane63(double v) {        // same network with fmax (inputs >= 0: the 0 a masked row reads is neutral)
    v = fmax(v, d_dpp_mov<0xB1>(v));
    v = fmax(v, d_dpp_mov<0x4E>(v));
    v = fmax(v, d_dpp_mov<0x141>(v));
    v = fmax(v, d_dpp_mov<0x140>(v));
    v = fmax(v, d_dpp_mov<0x142, 0xa>(v));
    v = fmax(v, d_dpp_mov<0x143, 0xc>(v));
    return v;
}

// Sum and max over a workgroup of NT threads in one pass: DPP inside the wave, one LDS slot per wave, the wave
// partials added in wave order by every thread (fixed order).  scratch: 2 * NT / 64 doubles.  All threads must call.
template <int NT>
DEV void d_block_sum_max(double &sum, double &mx, double *scratch, int tid) {
    const double s = d_wave_sum_to_lane63(sum), m = d_wave_max_to_lane63(mx);
    if ((tid & 63) == 63) { scratch[tid >> 6] = s; scratch[NT / 64 + (tid >> 6)] = m; }
    __syncthreads();
    double ts = 0.0, tm = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { ts += scratch[w]; tm = fmax(tm, scratch[NT / 64 + w]); }
    __syncthreads();
    sum = ts; mx = tm;
}

// Two sums and a maximum in one pass, same fixed order.  scratch: 3 * NT / 64 doubles.  All threads must call.
template <int NT>
DEV void d_block_sum2_max(double &a, double &b, double &mx, double *scratch, int tid) {
    const double sa = d_wave_sum_to_lane63(a), sb = d_wave_sum_to_lane63(b), m = d_wave_max_to_lane63(mx);
    if ((tid & 63) == 63) { scratch[tid >> 6] = sa; scratch[NT / 64 + (tid >> 6)] = sb; scratch[2 * (NT / 64) + (tid >> 6)] = m; }
    __syncthreads();
    double ta = 0.0, tb = 0.0, tm = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { ta += scratch[w]; tb += scratch[NT / 64 + w]; tm = fmax(tm, scratch[2 * (NT / 64) + w]); }
    __syncthreads();
    a = ta; b = tb; mx = tm;
}

// Two sums over a workgroup of NT threads in one pass (DPP inside the wave, wave partials added in wave order).
// scratch: 2 * NT / 64 doubles.  All threads must call; every thread gets both totals.
template <int NT>
DEV void d_block_sum2(double &a, double &b, double *scratch, int tid) {
    const double sa = d_wave_sum_to_lane63(a), sb = d_wave_sum_to_lane63(b);
    if ((tid & 63) == 63) { scratch[tid >> 6] = sa; scratch[NT / 64 + (tid >> 6)] = sb; }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { ta += scratch[w]; tb += scratch[NT / 64 + w]; }
    __syncthreads();
    a = ta; b = tb;
}

// deterministic block reductions (fixed tree), all threads must call
template <int NT>
DEV double d_block_sum(double v, double *scratch, int tid) {
    scratch[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (tid < s) scratch[tid] += scratch[tid + s];
        __syncthreads();
    }
    const double r = scratch[0];
    __syncthreads();
    return r;
}
template <int NT>
DEV double d_block_max(double v, double *scratch, int tid) {
    scratch[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (tid < s) scratch[tid] = fmax(scratch[tid], scratch[tid + s]);
        __syncthreads();
    }
    const double r = scratch[0];
    __syncthreads();
    return r;
}

#endif
