"""Writes tests/golden/lw_fsck_synth16.npz from the CPU oracle (oracle/ecckd_oracle.c), which is
itself pinned bit-for-bit by the reference-run values in kat_survey.json.  The reference's own
Fortran cannot be built or run in this environment (it needs RTE-RRTMGP), so these vectors are
oracle outputs, not reference outputs -- DESIGN.md "Oracle" says what that means for parity.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle  # noqa: E402
from rte_ecckd_amd import synthetic  # noqa: E402

m = oracle.CkdModel(os.path.join(ROOT, "data", "ecckd-1.2_lw_ckd-definition_climate_fsck-tol0.0161.nc"))
cols = synthetic.columns(0, 16, float(np.exp(m.log_pressure[0])))
tau, lay, inc, dec, sfc, err = oracle.gas_optics_int(m, cols["plev"], cols["tlay"], cols["tsfc"],
                                                     synthetic.gas_items(cols), cols["tlev"])
assert err == ""
fu, fd = oracle.rte_lw(tau, lay, inc, dec, np.repeat(cols["sfc_emis"][None], m.ng, 0), sfc)
np.savez_compressed(os.path.join(HERE, "lw_fsck_synth16.npz"), tau=tau, lay_source=lay, lev_source_inc=inc,
                    sfc_source=sfc, flux_up=fu, flux_dn=fd)
print("wrote lw_fsck_synth16.npz", tau.shape, float(fu[0].mean()), float(fd[-1].mean()))
