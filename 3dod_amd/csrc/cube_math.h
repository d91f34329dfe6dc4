// Device helpers shared by the fused 3D-head kernels (cube_head.hip: supervised losses and inference decode; weak_loss.hip: the
// weakly supervised losses): cuboid corners in the get_cuboid_verts_faces order (cubercnn/util/math_util.py:142-245 of the
// reference) with their back-propagation, and the allocentric -> egocentric ray rotation (math_util.py:802-830).
#pragma once
#include <math.h>

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }

// corners P[v][a] = sum_b R[a][b]*loc[v][b] + c[a];  loc = (sx*l/2, sy*h/2, sz*w/2), dims = (w,h,l)
__device__ __forceinline__ void loc_of(int v, const float* dims, float* loc) {
    loc[0] = (((v & 3) == 1 || (v & 3) == 2) ? 0.5f : -0.5f) * dims[2];
    loc[1] = ((v & 2) ? 0.5f : -0.5f) * dims[1];
    loc[2] = ((v & 4) ? 0.5f : -0.5f) * dims[0];
}
__device__ __forceinline__ void corners(const float* c, const float* dims, const float* R, float P[8][3]) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        float l[3];
        loc_of(v, dims, l);
#pragma unroll
        for (int a = 0; a < 3; ++a) P[v][a] = (R[a * 3] * l[0] + R[a * 3 + 1] * l[1] + R[a * 3 + 2] * l[2]) + c[a];
    }
}
// back-propagate dP (8x3) through corners(): accumulates dc[3], ddims[3] (w,h,l), dR[9]
__device__ __forceinline__ void corners_bwd(const float dP[8][3], const float* dims, const float* R, float* dc,
                                            float* ddims, float* dR) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        float l[3];
        loc_of(v, dims, l);
        float dl[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (dc) dc[a] += dP[v][a];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                if (dR) dR[a * 3 + b] += dP[v][a] * l[b];
                dl[b] += R[a * 3 + b] * dP[v][a];
            }
        }
        if (ddims) {
            ddims[2] += (((v & 3) == 1 || (v & 3) == 2) ? 0.5f : -0.5f) * dl[0];
            ddims[1] += ((v & 2) ? 0.5f : -0.5f) * dl[1];
            ddims[0] += ((v & 4) ? 0.5f : -0.5f) * dl[2];
        }
    }
}
// rotation taking the optical axis to the viewing ray through (uu,vv): pytorch3d axis_angle_to_matrix semantics
__device__ __forceinline__ bool ray_rotation(float uu, float vv, const float* K4, float* M) {
    float ox = (uu - K4[2]) / K4[0], oy = (vv - K4[3]) / K4[1], oz = 1.f;
    const float nrm = sqrtf((ox * ox + oy * oy) + oz * oz);
    ox /= nrm; oy /= nrm; oz /= nrm;
    const float angle = acosf(oz);
    const float an = sqrtf(oy * oy + ox * ox);
    const float ax = angle * (-oy) / an, ay = angle * ox / an, az = 0.f;
    const float ang = sqrtf((ax * ax + ay * ay) + az * az);
    const float half = ang * 0.5f;
    const float s = fabsf(ang) < 1e-6f ? 0.5f - (ang * ang) / 48.f : sinf(half) / ang;
    const float qr = cosf(half), qi = ax * s, qj = ay * s, qk = az * s;
    const float two_s = 2.0f / (((qr * qr + qi * qi) + qj * qj) + qk * qk);
    M[0] = 1 - two_s * (qj * qj + qk * qk); M[1] = two_s * (qi * qj - qk * qr); M[2] = two_s * (qi * qk + qj * qr);
    M[3] = two_s * (qi * qj + qk * qr); M[4] = 1 - two_s * (qi * qi + qk * qk); M[5] = two_s * (qj * qk - qi * qr);
    M[6] = two_s * (qi * qk - qj * qr); M[7] = two_s * (qj * qk + qi * qr); M[8] = 1 - two_s * (qi * qi + qj * qj);
    return angle > 0.f;
}

