// placeholder until the MFMA family lands: nothing is "supported", the generic family runs.
#include "d3pm_kernels.h"
namespace d3pm {
bool mfma_linear_supported(int, const LinearArgs&) { return false; }
int mfma_linear(int, const LinearArgs&, hipStream_t) { return D3PM_E_SHAPE; }
bool mfma_attention_supported(int, const AttnArgs&) { return false; }
int mfma_attention(int, const AttnArgs&, hipStream_t) { return D3PM_E_SHAPE; }
}  // namespace d3pm
