/* evc_oracle_priv.h — CPU ORACLE (test infrastructure): private network struct. */
#ifndef EVC_ORACLE_PRIV_H
#define EVC_ORACLE_PRIV_H
#include "evc_oracle.h"
struct orc_net {
    int n, m;
    double A[ORC_MAX_CONSTRAINTS * ORC_MAX_STATIONS];          /* cn.constraint_matrix */
    double phase_deg[ORC_MAX_STATIONS];                        /* cn._phase_angles     */
    double cosphi[ORC_MAX_STATIONS], sinphi[ORC_MAX_STATIONS]; /* np.exp(1j*np.deg2rad(.)) env.py:485 */
    double mag[ORC_MAX_CONSTRAINTS];                           /* cn.magnitudes        */
    uint8_t kind[ORC_MAX_STATIONS];                            /* 0 AV, 1 CC           */
};
#endif
