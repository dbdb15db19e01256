// vg_scan_ex.hip - the EX = true instantiations of vg_scan_kernel (vg_scan.h): the plain scan with what tie_order = reference
// needs on top - a start threshold from the pass over the rows in front, the candidate stream, "top-k + store" for the prefix pass
// (vg_reforder.hip).  A translation unit of their own: compile time, and the plain kernels keep their register budget.
// One load policy (non-temporal): the prefix pass is small and the emitting main pass runs only while ties are around.
#include "vg_internal.h"

#include "vg_scan.h"

typedef void (*scan_fn_t)(ScanArgs);

template <int VT, int ACC>
static scan_fn_t pick_u(int U) {
    switch (U) {
        case 1: return vg_scan_kernel<VT, ACC, 1, true, true>;
        case 2: return vg_scan_kernel<VT, ACC, 2, true, true>;
        case 3: return vg_scan_kernel<VT, ACC, 3, true, true>;
        case 4: return vg_scan_kernel<VT, ACC, 4, true, true>;
        case 6: return vg_scan_kernel<VT, ACC, 6, true, true>;
        case 8: return vg_scan_kernel<VT, ACC, 8, true, true>;
    }
    return nullptr;
}

template <int VT>
static scan_fn_t pick_acc(int acc, int U) {
    switch (acc) {
        case A_L2: return pick_u<VT, A_L2>(U);
        case A_COS: return pick_u<VT, A_COS>(U);
        case A_DOT: return pick_u<VT, A_DOT>(U);
        case A_L1: return pick_u<VT, A_L1>(U);
        case A_COSN:
            if constexpr (VT == T_F16 || VT == T_BF16) return pick_u<VT, A_COSN>(U);
            return nullptr;
    }
    return nullptr;
}

// the EX kernel for (element type, accumulation kind, chunks per lane), nullptr if there is none
void (*vg_pick_scan_kernel_ex(int vtype, int acc, int U))(ScanArgs) {
    switch (vtype) {
        case VG_TYPE_F32: return pick_acc<T_F32>(acc, U);
        case VG_TYPE_U8: return pick_acc<T_U8>(acc, U);
        case VG_TYPE_I8: return pick_acc<T_I8>(acc, U);
        case VG_TYPE_F16: return pick_acc<T_F16>(acc, U);
        case VG_TYPE_BF16: return pick_acc<T_BF16>(acc, U);
    }
    return nullptr;
}
