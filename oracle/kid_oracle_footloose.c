/* placeholder, replaced below */
#include "kid_oracle.h"
void ko_footloose_calving(const ko_grid *g, const kid_params *p, kid_berg_soa *b, int64_t capacity, double *acc, double *scalars) {
  (void)g; (void)p; (void)b; (void)capacity; (void)acc; (void)scalars;
}
