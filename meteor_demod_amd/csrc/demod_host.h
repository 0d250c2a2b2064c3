/*
 * demod_host.h — host-side (init-time) derivations of the demodulator:
 * demod_init's parameter arithmetic, the RRC polyphase table and the tanh LUT,
 * computed with the host libm exactly as the reference does at start-up
 * (CPU and GPU paths must share bit-identical tables).
 */
#ifndef MDEMOD_HOST_H
#define MDEMOD_HOST_H

#include <vector>
#include "demod_internal.h"

struct HostTables {
	DemodConsts        c;
	float              osf;
	std::vector<float> rrc;       /* interp*taps, bank-major (filter.c:20)     */
	std::vector<float> ctab;      /* [4 alignments][interp banks][row stride]  */
	float              tanh_lut[32];
	bool               rw_wide;   /* wide geometry (160-slot packed window, compact table, 512-thread blocks): <= 129 taps at <= 15 samples per firing, on v3 also 66..129 taps at 15..30 (wide_far_ok) */
	bool               rw_far;    /* far geometry: <= 65 taps at 15..46 samples per firing (112-slot packed window, 47 alignments; far_ok); with rw_hyb: the 120-slot hybrid window, float input at 30..54 (hyb_far_ok) */
	bool               rw_mid;    /* mid geometry: <= 65 taps at 3.6..15 samples per firing (96-slot packed window, compact table); with rw_hyb: the 96-slot hybrid window, float input, <= 65 taps at <= 30 */
	bool               rw_std_compact; /* std geometry on the v3 kernel with the compact4 coefficient table (large -O) */
	bool               rw_gather; /* v3 gather geometry: s16 input at rates beyond every window (demod_kernel_gat.hip) */
	bool               rw_hyb;    /* v3 hybrid window, float input outside the std geometry (hyb_ok): <= 129 taps at <= 30 samples per firing (160 slots: 80 in VGPRs + 80 in AccVGPRs); with rw_mid <= 65 taps at <= 30 (96 slots); with rw_far <= 65 taps at 30..54 (120 slots, 55 alignments); compact4 table */
	bool               rw_compact4; /* wide / mid / far on the v3 packed rotating window (demod_kernel_rotp.hip): four shifted copies per bank, 16-byte reads */
	bool               use_rw;    /* one of the v3 register-window kernels serves this configuration (std / wide / mid / far / hybrid / gather) */
};

/* Returns MDEMOD_OK or MDEMOD_ERR_PARAM. */
/* generation: 0 = v1 LDS ring only (tests), anything else = the v3 rotating windows where they apply (default) */
int mdemod_host_derive(const mdemod_params &p, HostTables &out, int generation = 2);

#endif
