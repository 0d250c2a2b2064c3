"""Run the C host's full-screen display and return what it drew: on a pseudo-terminal (where it starts by itself), or on pipes
with --tui (boxes without /dev/pts)."""
from __future__ import annotations

import fcntl
import os
import pty
import re
import select
import struct
import subprocess
import termios
import time


def run_in_pty(argv, script=((b"Press any key", b"x"),), rows=30, cols=100, timeout=60.0):
    """-> (return code, screen text with the escape sequences taken out, stderr).  `script`: (text, keys) pairs - the keys are
    typed once the text has been drawn, one pair after the other."""
    master, slave = pty.openpty()            # OSError where there are no pseudo-terminals
    fcntl.ioctl(slave, termios.TIOCSWINSZ, struct.pack("HHHH", rows, cols, 0, 0))
    proc = subprocess.Popen(argv, stdin=slave, stdout=slave, stderr=subprocess.PIPE, env=dict(os.environ, TERM="xterm"), close_fds=True)
    os.close(slave)
    return _drive(proc, master, master, script, timeout)


def run_on_pipes(argv, script=((b"Press any key", b"x"),), rows=30, cols=100, timeout=60.0):
    """The same over plain pipes: argv must carry --tui; the screen size comes from LINES / COLUMNS."""
    env = dict(os.environ, TERM="xterm", LINES=str(rows), COLUMNS=str(cols))
    proc = subprocess.Popen(argv, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, close_fds=True)
    return _drive(proc, proc.stdout.fileno(), proc.stdin.fileno(), script, timeout, close=False)


def _drive(proc, rfd, wfd, script, timeout, close=True):
    master = rfd
    out, step, t0 = b"", 0, time.time()
    while time.time() - t0 < timeout:
        ready, _, _ = select.select([master], [], [], 0.2)
        if ready:
            try:
                chunk = os.read(master, 65536)
            except OSError:          # the child closed its end
                break
            if not chunk:
                break
            out += chunk
        if step < len(script) and script[step][0] in out:
            os.write(wfd, script[step][1])
            step += 1
        if proc.poll() is not None and not ready:
            break
    try:
        proc.wait(timeout=10)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.wait()
    if close:
        os.close(master)
    text = re.sub(rb"\x1b\[[0-9;?]*[A-Za-z]|\x1b\([A-Z0-9]|\x1b[=>]", b" ", out).decode("latin1")
    return proc.returncode, text, proc.stderr.read().decode("latin1")
