// Real spherical-harmonic basis, degrees 0-3, in the reference's sign/ordering convention
// (constants: GR/compact.cu:553-570 == GR/transform.cu:932-949; polynomials: GR/compact.cu:574-653).
// rgb = sum_k b[k] * sh[k] + 0.5 with sh[0] = sh_0 and sh[k>=1] = sh_rest[k-1].
#pragma once

template <int DEG>
__device__ __forceinline__ void lg_sh_basis(float x, float y, float z, float* b)
{
    constexpr float C0 = 0.28209479177387814f;
    constexpr float C1 = 0.4886025119029199f;
    b[0] = C0;
    if (DEG > 0) {
        b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
    }
    if (DEG > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = 1.0925484305920792f * xy;
        b[5] = -1.0925484305920792f * yz;
        b[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
        b[7] = -1.0925484305920792f * xz;
        b[8] = 0.5462742152960396f * (xx - yy);
        if (DEG > 2) {
            b[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
            b[10] = 2.890611442640554f * xy * z;
            b[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
            b[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            b[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
            b[14] = 1.445305721320277f * z * (xx - yy);
            b[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
        }
    }
}
