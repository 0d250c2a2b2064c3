/*
 * tui.c — full-screen display of the C host, see tui.h.  Layout (recomputed on every terminal resize):
 *
 *   row 0        title
 *   rows 2..     left: the constellation, a square-looking plot (twice as wide as high, at most 31 columns: tui.h:9 of the
 *                reference); right: "PLL status" (3 rows), "Data in" (2), "Data out" (2)
 *   below        the message log, scrolling
 *
 * The demodulation loop owns the pace: tui_draw() never blocks (the reference's tui_process_input() sleeps for the refresh
 * period instead, because its demodulator runs in another thread: main.c:224-229).
 */
#include "tui.h"

#include <curses.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#define PLOT_MAX_COLS 31

static struct {
	int     up;
	SCREEN *term;
	WINDOW *title, *plot, *pll, *in, *out, *log;
	int     plot_rows, plot_cols;
} scr;

/* ---- terminal-free helpers ------------------------------------------------------------------------------------------- */

void
tui_fmt_size(unsigned long n, char out[16])
{
	static const char unit[] = " kMGTPE";
	float v = (float)n;
	int u = 0;
	if (n >= 1000)
		while (v > 1000 && u < 6) { v /= 1000; u++; }
	if (n < 1000) snprintf(out, 16, "%lu %c", n, unit[0]);
	else snprintf(out, 16, "%3.*f %c", v > 99.9f ? 0 : v > 9.99f ? 1 : 2, v, unit[u]);
}

void
tui_fmt_clock(unsigned long seconds, char out[16])
{
	if (seconds > 99ul * 3600) seconds = 0;
	snprintf(out, 16, "%02lu:%02lu:%02lu", seconds / 3600, seconds / 60 % 60, seconds % 60);
}

void
tui_constellation(const int8_t *iq, unsigned n_symbols, int rows, int cols, unsigned char *hits)
{
	memset(hits, 0, (size_t)rows * (size_t)cols);
	for (unsigned k = 0; k < n_symbols; k++) {
		const int c = cols / 2 + iq[2 * k] * cols / 255;
		const int r = rows / 2 - iq[2 * k + 1] * rows / 255;
		if (r < 0 || r >= rows || c < 0 || c >= cols) continue;
		unsigned char *h = &hits[r * cols + c];
		if (*h < 4) (*h)++;
	}
}

char
tui_glyph(unsigned char hits)
{
	return " .-+#"[hits > 4 ? 4 : hits];
}

/* ---- the screen -------------------------------------------------------------------------------------------------------- */

static void
drop(WINDOW **w)
{
	if (*w) delwin(*w);
	*w = NULL;
}

static void
layout(void)
{
	int rows, cols;
	getmaxyx(stdscr, rows, cols);
	WINDOW *old_log = scr.log;
	drop(&scr.title); drop(&scr.plot); drop(&scr.pll); drop(&scr.in); drop(&scr.out);
	scr.plot_cols = cols / 2 < PLOT_MAX_COLS ? (cols / 2) | 1 : PLOT_MAX_COLS;
	scr.plot_rows = scr.plot_cols / 2;
	const int right = scr.plot_cols + 2, right_w = cols - right;
	int log_top = 2 + scr.plot_rows + 1;
	if (log_top < 12) log_top = 12;
	erase();
	refresh();
	scr.title = newwin(1, cols, 0, 0);
	if (rows > 2 + scr.plot_rows) scr.plot = newwin(scr.plot_rows, scr.plot_cols, 2, 0);
	if (right_w > 8) {
		if (rows > 5) scr.pll = newwin(3, right_w, 2, right);
		if (rows > 8) scr.in = newwin(2, right_w, 6, right);
		if (rows > 11) scr.out = newwin(2, right_w, 9, right);
	}
	if (rows > log_top + 1) {
		scr.log = newwin(rows - log_top, cols, log_top, 0);
		if (scr.log) {
			scrollok(scr.log, TRUE);
			keypad(scr.log, TRUE);
			nodelay(scr.log, TRUE);
			if (old_log) overwrite(old_log, scr.log);
		}
	} else scr.log = NULL;
	if (old_log) delwin(old_log);
	if (scr.title) {
		wattron(scr.title, A_BOLD);
		mvwprintw(scr.title, 0, 2, "~ meteor_demod_amd: Meteor-M LRPT demodulator on MI355X ~");
		wattroff(scr.title, A_BOLD);
		wrefresh(scr.title);
	}
}

int
tui_open(int refresh_ms)
{
	(void)refresh_ms;
	if (scr.up) return 0;
	/* newterm, not initscr: an unknown or missing TERM makes initscr() exit the program; here the caller falls back to status lines */
	scr.term = newterm(NULL, stdout, stdin);
	if (!scr.term) return 1;
	set_term(scr.term);
	cbreak();
	noecho();
	curs_set(0);
	keypad(stdscr, TRUE);
	nodelay(stdscr, TRUE);
	scr.up = 1;
	layout();
	return 0;
}

void
tui_close(void)
{
	if (!scr.up) return;
	drop(&scr.title); drop(&scr.plot); drop(&scr.pll); drop(&scr.in); drop(&scr.out); drop(&scr.log);
	endwin();
	if (scr.term) delscreen(scr.term);
	scr.term = NULL;
	scr.up = 0;
}

int
tui_log(const char *fmt, ...)
{
	if (!scr.up || !scr.log) return 0;
	char stamp[16];
	const time_t now = time(NULL);
	strftime(stamp, sizeof(stamp), "%H:%M:%S", localtime(&now));
	wprintw(scr.log, "(%s) ", stamp);
	va_list ap;
	va_start(ap, fmt);
	vw_printw(scr.log, fmt, ap);
	va_end(ap);
	wrefresh(scr.log);
	return 0;
}

static void
heading(WINDOW *w, const char *text)
{
	werase(w);
	wattron(w, A_BOLD);
	mvwprintw(w, 0, 0, "%s", text);
	wattroff(w, A_BOLD);
}

int
tui_draw(const struct tui_frame *f)
{
	if (!scr.up) return 0;
	/* keys first: q leaves, a resize rebuilds the panes */
	int quit = 0;
	for (;;) {
		const int key = scr.log ? wgetch(scr.log) : getch();
		if (key == ERR) break;
		if (key == KEY_RESIZE) layout();
		else if (key == 'q' || key == 'Q') quit = 1;
	}
	if (scr.plot) {
		unsigned char hits[PLOT_MAX_COLS * PLOT_MAX_COLS];
		const int nr = scr.plot_rows, nc = scr.plot_cols;
		tui_constellation(f->symbols, f->symbols ? f->n_symbols : 0, nr, nc, hits);
		werase(scr.plot);
		for (int r = 0; r < nr; r++)
			for (int c = 0; c < nc; c++) {
				const unsigned char h = hits[r * nc + c];
				if (h) mvwaddch(scr.plot, r, c, (chtype)tui_glyph(h));
				else if (r == nr / 2 && c == nc / 2) mvwaddch(scr.plot, r, c, ACS_PLUS);
				else if (r == nr / 2) mvwaddch(scr.plot, r, c, ACS_HLINE);
				else if (c == nc / 2) mvwaddch(scr.plot, r, c, ACS_VLINE);
			}
		wrefresh(scr.plot);
	}
	if (scr.pll) {
		heading(scr.pll, "PLL status: ");
		wattron(scr.pll, A_BOLD);
		wprintw(scr.pll, f->locked ? "Locked\n" : "Acquiring...\n");
		wattroff(scr.pll, A_BOLD);
		wprintw(scr.pll, "Gain\tCarrier freq\tSymbol rate\n");
		wprintw(scr.pll, "%.3f\t%+7.1f Hz\t%7.1f Hz", f->gain, f->carrier_hz, f->symrate_hz);
		wrefresh(scr.pll);
	}
	if (scr.in) {
		char done[16], total[16];
		const unsigned bps = f->in_bytes_per_second ? f->in_bytes_per_second : 1;
		tui_fmt_clock(f->in_done / bps, done);
		tui_fmt_clock(f->in_total / bps, total);
		heading(scr.in, "Data in");
		mvwprintw(scr.in, 1, 0, "%s/%s (%.1f%%)", done, total, f->in_total ? 100.0 * (double)f->in_done / (double)f->in_total : 0.0);
		wrefresh(scr.in);
	}
	if (scr.out) {
		char size[16];
		tui_fmt_size(f->out_bytes, size);
		heading(scr.out, "Data out");
		mvwprintw(scr.out, 1, 0, "%sB", size);
		wrefresh(scr.out);
	}
	return quit;
}

int
tui_wait_key(void)
{
	if (!scr.up) return 0;
	WINDOW *w = scr.log ? scr.log : stdscr;
	nodelay(w, FALSE);
	int key;
	do key = wgetch(w); while (key == KEY_RESIZE);
	nodelay(w, TRUE);
	return key;
}
