"""The host program's TCP sinks (-s connect, -l listen: the reference's outmode 1 / 2, main.c:65-72, output.c:59-157,
318-336) without a GPU: tests/cpp/sink_harness.c drives adsbdec_amd/csrc/cli/sink.c with packets from the library's own
formatter over loopback sockets.  The end-to-end run (the real program decoding a capture into a socket, compared with
the golden packets, which were minted through the reference's own formatpkt) is tests/test_gpu_parity.py::test_cli_tcp_sinks_carry_the_same_packets."""
import os
import socket
import subprocess
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI_DIR = os.path.join(ROOT, "adsbdec_amd", "csrc", "cli")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sink") / "sink_harness")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-o", out, os.path.join(ROOT, "tests", "cpp", "sink_harness.c"),
                    os.path.join(CLI_DIR, "sink.c"), os.path.join(ROOT, "adsbdec_amd", "csrc", "format.c"),
                    "-I", os.path.join(ROOT, "include"), "-lm"], check=True)
    return out


def _free_port(family=socket.AF_INET, host="127.0.0.1"):
    with socket.socket(family, socket.SOCK_STREAM) as s:
        s.bind((host, 0))
        return s.getsockname()[1]


class Listener:
    """Accepts one peer and reads until it closes (or until `stop_after` bytes, then closes on its side)."""

    def __init__(self, family=socket.AF_INET, host="127.0.0.1", stop_after=None, port=0):
        self.sock = socket.socket(family, socket.SOCK_STREAM)
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.sock.bind((host, port))
        self.sock.listen(1)
        self.port = self.sock.getsockname()[1]
        self.data = b""
        self.stop_after = stop_after
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        conn, _ = self.sock.accept()
        with conn:
            while True:
                b = conn.recv(1 << 16)
                if not b:
                    break
                self.data += b
                if self.stop_after is not None and len(self.data) >= self.stop_after:
                    break
        self.sock.close()

    def join(self):
        self.t.join(30)
        assert not self.t.is_alive()
        return self.data


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_connect_mode_sends_every_byte_in_order(harness, tmp_path, fmt):
    """-s: 20 000 packets (several 64 KiB batches; Beast with its doubled 0x1a bytes) arrive byte for byte."""
    lis = Listener()
    copy = str(tmp_path / "copy.bin")
    p = subprocess.run([harness, "1", f"127.0.0.1:{lis.port}", str(fmt), "20000", copy], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    got = lis.join()
    want = open(copy, "rb").read()
    assert got == want and len(want) > 300_000
    assert p.stderr.splitlines()[0] == "connected"
    assert "lost 0, packets dropped 0" in p.stderr


def test_listen_mode_accepts_one_peer(harness, tmp_path):
    """-l: "listening" then "connected" on stderr (output.c:135,145), one accepted peer gets every packet, the listening
    socket is gone afterwards (a second connection is refused)."""
    port = _free_port()
    copy = str(tmp_path / "copy.bin")
    p = subprocess.Popen([harness, "2", f"127.0.0.1:{port}", "1", "5000", copy, "20"], stderr=subprocess.PIPE, text=True)
    assert p.stderr.readline().strip() == "listening"
    c = socket.create_connection(("127.0.0.1", port), timeout=10)
    assert p.stderr.readline().strip() == "connected"
    with pytest.raises(OSError):
        socket.create_connection(("127.0.0.1", port), timeout=2).close()
    got = b""
    while True:
        b = c.recv(1 << 16)
        if not b:
            break
        got += b
    c.close()
    assert p.wait(30) == 0
    assert got == open(copy, "rb").read() and got.count(b"\n") == 5000


def test_ipv6_bracket_form(harness, tmp_path):
    if not socket.has_ipv6:
        pytest.skip("no IPv6 here")
    try:
        lis = Listener(socket.AF_INET6, "::1")
    except OSError:
        pytest.skip("no IPv6 loopback here")
    copy = str(tmp_path / "copy.bin")
    p = subprocess.run([harness, "1", f"[::1]:{lis.port}", "0", "300", copy], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    assert lis.join() == open(copy, "rb").read()


def test_default_ports(harness, tmp_path):
    """No port in the address: 30001 for -s, 30002 for -l (output.c:84,93)."""
    copy = str(tmp_path / "copy.bin")
    try:
        lis = Listener(port=30001)
    except OSError:
        pytest.skip("port 30001 is taken on this machine")
    p = subprocess.run([harness, "1", "127.0.0.1", "0", "10", copy], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and lis.join() == open(copy, "rb").read()
    p = subprocess.Popen([harness, "2", "127.0.0.1", "0", "10", copy], stderr=subprocess.PIPE, text=True)
    line = p.stderr.readline().strip()
    if line != "listening":
        p.kill()
        pytest.skip("port 30002 is taken on this machine")
    c = socket.create_connection(("127.0.0.1", 30002), timeout=10)
    got = b""
    while True:
        b = c.recv(4096)
        if not b:
            break
        got += b
    assert p.wait(30) == 0 and got == open(copy, "rb").read()


def test_unusable_addresses_end_the_run(harness, tmp_path):
    """runOutput() returns -1 -> exit status 255, with the reference's messages (output.c:77,103)."""
    copy = str(tmp_path / "copy.bin")
    p = subprocess.run([harness, "1", "[::1", "0", "1", copy], capture_output=True, text=True, timeout=60)
    assert p.returncode == 255 and p.stderr == "Invalid IPV6 address\n"
    p = subprocess.run([harness, "1", "no.such.host.invalid:5", "0", "1", copy], capture_output=True, text=True, timeout=60)
    assert p.returncode == 255 and p.stderr == "Invalid/unknown address no.such.host.invalid\n"


def test_waits_for_a_peer_that_is_not_there_yet(harness, tmp_path):
    """-s before anybody listens: one attempt every retry_s seconds (3 in the reference, output.c:282) until a peer is there."""
    port = _free_port()
    copy = str(tmp_path / "copy.bin")
    env = dict(os.environ, ADSB_CLI_RETRY_S="1")
    p = subprocess.Popen([harness, "1", f"127.0.0.1:{port}", "0", "1000", copy], stderr=subprocess.PIPE, text=True, env=env)
    time.sleep(1.5)
    assert p.poll() is None                      # still trying
    lis = Listener(port=port)
    assert p.wait(30) == 0
    assert lis.join() == open(copy, "rb").read()


def test_a_peer_that_goes_away(harness, tmp_path):
    """The peer closes after the first bytes: "disconnected" on stderr, the batch in flight is dropped, later batches
    try once for a new peer and are dropped without one; the run ends with status 0 (a lost peer is not an error,
    output.c:321-327).  What the peer did receive is a prefix of what was sent."""
    lis = Listener(stop_after=1)
    copy = str(tmp_path / "copy.bin")
    p = subprocess.run([harness, "1", f"127.0.0.1:{lis.port}", "0", "60000", copy, "30"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    got = lis.join()
    lines = p.stderr.splitlines()
    assert lines[0] == "connected" and "disconnected" in lines
    sent = open(copy, "rb").read()               # the batches the sink reported as written
    assert len(got) > 0 and sent[:len(got)] == got[:len(sent)]
    lost = int(lines[-1].split("lost ")[1].split(",")[0])
    assert lost >= 1 and "packets dropped 0" not in lines[-1]


def test_a_listening_sink_whose_peer_goes_away_does_not_wait_for_another(harness, tmp_path):
    """-l: the one accepted peer closes early.  The reference would sit in accept() again (output.c:139) until somebody
    connects; this program drops the rest of the run's packets and ends."""
    port = _free_port()
    copy = str(tmp_path / "copy.bin")
    p = subprocess.Popen([harness, "2", f"127.0.0.1:{port}", "0", "60000", copy, "30"], stderr=subprocess.PIPE, text=True)
    assert p.stderr.readline().strip() == "listening"
    c = socket.create_connection(("127.0.0.1", port), timeout=10)
    assert p.stderr.readline().strip() == "connected"
    assert len(c.recv(1)) == 1
    c.close()
    assert p.wait(60) == 0
    rest = p.stderr.read().splitlines()
    assert "disconnected" in rest and rest.count("listening") == 0
    assert "packets dropped 0" not in rest[-1]


def test_the_host_program_refuses_unusable_addresses_before_it_touches_the_gpu(tmp_path):
    """adsbdec_amd_cli itself (built by __graft_entry__.build()): -s / -l with an address that cannot be used end the run with
    status 255 and the reference's message before adsb_create is reached -- so this runs on a box without a GPU; -s together
    with several captures, and unknown options, print the usage text and exit 1 (main.c:85-87)."""
    from adsbdec_amd import capi
    if not os.path.exists(capi.CLI_PATH):
        pytest.skip("the host program is not built")
    f = tmp_path / "z.u16"
    f.write_bytes(b"\0" * 4096)
    p = subprocess.run([capi.CLI_PATH, "-s", "[::1", "-f", str(f)], capture_output=True, timeout=60)
    assert p.returncode == 255 and p.stderr == b"Invalid IPV6 address\n" and p.stdout == b""
    p = subprocess.run([capi.CLI_PATH, "-l", "no.such.host.invalid:1", "-f", str(f)], capture_output=True, timeout=60)
    assert p.returncode == 255 and p.stderr == b"Invalid/unknown address no.such.host.invalid\n"
    p = subprocess.run([capi.CLI_PATH, "-G", "0,0", "-s", "127.0.0.1:9", "-f", str(f), "-f", str(f)], capture_output=True, timeout=60)
    assert p.returncode == 1 and b"usage" in p.stdout
    p = subprocess.run([capi.CLI_PATH, "-s"], capture_output=True, timeout=60)      # option without its argument
    assert p.returncode == 1


# ---- the host program's signal behaviour (main.c:91-99), as far as it can be seen without a GPU ---------------------------
CLI = os.path.join(ROOT, "adsbdec_amd", "lib", "adsbdec_amd_cli")


@pytest.fixture(scope="module")
def cli():
    from adsbdec_amd import _build
    _build.build()
    assert os.path.exists(CLI)
    return CLI


@pytest.mark.parametrize("extra", [[], ["-G", "1"]])
@pytest.mark.parametrize("sig", ["SIGINT", "SIGTERM", "SIGQUIT"])
def test_a_signal_while_waiting_for_a_peer_ends_the_run_with_the_table(cli, tmp_path, sig, extra):
    """The reference sits in accept() (output.c:139) until a peer connects; SIGINT / SIGTERM / SIGQUIT (handlerExit,
    main.c:91-96, no SA_RESTART) end runOutput() with 0 and main prints the Try/Ok table (main.c:103).  The peer is waited
    for BEFORE the GPU runtime is started, so this much of the program runs anywhere."""
    import signal
    cap = tmp_path / "c.u16"
    cap.write_bytes(bytes(4096))
    port = _free_port()
    p = subprocess.Popen([cli] + extra + ["-l", f"127.0.0.1:{port}", "-f", str(cap)], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    assert p.stderr.readline().strip() == "listening"
    time.sleep(0.2)
    p.send_signal(getattr(signal, sig))
    out, err = p.communicate(timeout=20)
    assert p.returncode == 0, (p.returncode, err)
    lines = err.splitlines()
    assert [ln.split(":")[0].strip() for ln in lines[-3:]] == ["Try", "Ok", "Total"], err
    assert lines[-1].split()[-1] == "0" and out == ""


def test_more_f_arguments_than_the_program_keeps_are_refused_not_dropped(cli, tmp_path):
    args = []
    for k in range(65):
        args += ["-f", str(tmp_path / f"c{k}.u16")]
    p = subprocess.run([cli, "-G", "2"] + args, capture_output=True, text=True, timeout=20)
    assert p.returncode == 1 and "usage" in p.stdout


def test_a_failed_last_flush_is_an_error_not_a_silent_truncation(harness):
    """ENOSPC on the LAST buffered block only shows when stdout is flushed (sink_close): the status must say so."""
    with open("/dev/full", "wb") as full:
        few = subprocess.run([harness, "0", "x", "0", "10", "/dev/null"], stdout=full, stderr=subprocess.PIPE, timeout=60)
        many = subprocess.run([harness, "0", "x", "0", "5000", "/dev/null"], stdout=full, stderr=subprocess.PIPE, timeout=60)
    assert few.returncode != 0 and many.returncode != 0
