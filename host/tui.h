/*
 * tui.h — the full-screen display of the C host (ncurses): what the reference draws when it is not given -B
 * (tui.c / tui.h, main.c:197,224-245): a coarse constellation, the PLL's state, input progress, output size and a
 * message log.  Own implementation: one frame structure filled by the demodulation loop from the status snapshot
 * (include/meteor_demod_amd.h: mdemod_status) and drawn in one call.
 */
#ifndef MDEMOD_HOST_TUI_H
#define MDEMOD_HOST_TUI_H

#include <stdint.h>

/* what one redraw shows */
struct tui_frame {
	double        carrier_hz, symrate_hz, gain;   /* main.c:231-232,237: pll_get_freq, mm_omega in Hz, agc_get_gain */
	int           locked;                         /* pll_get_locked */
	unsigned long in_done, in_total;              /* bytes of the input file consumed / its length (0: unknown) */
	unsigned      in_bytes_per_second;            /* 2 * samplerate * bps / 8 (main.c:235) */
	unsigned long out_bytes;                      /* soft symbols written so far */
	const int8_t *symbols;                        /* the latest soft symbols, I Q I Q ... */
	unsigned      n_symbols;
};

int  tui_open(int refresh_ms);              /* 0 = the screen is up */
void tui_close(void);
int  tui_log(const char *fmt, ...);         /* printf-compatible: "(HH:MM:SS) message" into the log pane */
int  tui_draw(const struct tui_frame *f);   /* redraws every pane; 1 = the user pressed q */
int  tui_wait_key(void);                    /* blocks until a key is pressed */

/* ---- the parts that need no terminal (tested on their own: `meteor_demod_amd --tui-selftest`) ---- */

/* utils.c:22-41: "999 ", "1.23 k", "12.3 M", "123 G" (three significant digits, powers of 1000) */
void tui_fmt_size(unsigned long n, char out[16]);
/* utils.c:44-57: HH:MM:SS, "00:00:00" beyond 99 hours */
void tui_fmt_clock(unsigned long seconds, char out[16]);
/* tui.c:166-201: hits per character cell of a rows x cols plot with the origin in the middle; a soft value v lands in column
 * cols/2 + v*cols/255, row rows/2 - v*rows/255 (integer arithmetic); counts saturate at 4 (glyphs " .-+#") */
void tui_constellation(const int8_t *iq, unsigned n_symbols, int rows, int cols, unsigned char *hits);
char tui_glyph(unsigned char hits);

#endif
