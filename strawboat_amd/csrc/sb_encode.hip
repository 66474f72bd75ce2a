#include "sb_host.h"
extern "C" {
uint64_t sb_write_bound(int32_t, int32_t, uint64_t, uint64_t, const sb_write_options*, uint64_t* n_pages) {
    if (n_pages) *n_pages = 0;
    return 0;
}
int32_t sb_write_columns(sb_ctx* ctx, sb_column_write*, uint64_t, const sb_write_options*, int32_t) {
    return ctx ? ctx->fail(SB_ERR_NYI, "encode not built yet") : SB_ERR_INVALID;
}
}
