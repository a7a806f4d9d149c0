"""Replay of the reference test flow without a simulator -- test infrastructure.

Reads a configuration of the shape the reference persists as tb/tmp/<seed>.json
(config/gcm_utils.py:248-263; keys seed, key, iv, aad, data, enc_dec, aes_mode, max_n_byte, ...), makes the
same draws tb/gcm_gctr.py makes, in the same order -- config_data (:229-332: IV and key as hex strings,
n_bytes = int(betavariate(.1, .1) * max_n_byte) for AAD and data, 5 delay bits) and encrypt_data (:340-415:
full beats randint(0, 2^128-1), last beat a random value of the remaining 1..15 bytes) -- and drives a model
object exactly as tb/gcm_test.py:45,76-94 does: load_aad per AAD beat, load_plain_text / load_cipher_text per
data beat, get_tag with the DUT's tag, then compares the scoreboard lists.

The RTL DUT is replaced by a stand-in (`dut`: a callable (key, iv, aad, data, dec) -> (out, tag)); with the
CPU oracle as the DUT this checks the GPU-backed drop-in class end to end under the reference's own stimulus.
Whether the draws are bit-identical to a real cocotb run (cocotb seeds `random` with RANDOM_SEED before the
test starts) cannot be verified here: cocotb and GHDL are not installable in this environment.
"""
import random
import re

RANDOM_PARAM, EMPTY_PARAM = 'RANDOM', 'EMPTY'         # tb/gcm_gctr.py:19-20


def default_config(seed, aes_mode='128', enc_dec='enc', test_size='short', **kw):
    """What config/gcm_utils.py test_config()/gcm_ip_config() would persist for `python gcm_testbench.py`."""
    max_n = {'short': 2 ** 12 - 1, 'medium': 2 ** 16 - 1, 'long': 2 ** 32 - 1}[test_size]     # gcm_utils.py:144
    cfg = dict(seed=seed, key=RANDOM_PARAM, test_size=test_size, iv=RANDOM_PARAM, aad=RANDOM_PARAM, data=RANDOM_PARAM,
               enc_dec=enc_dec, compiler='ghdl', max_n_byte=max_n, aes_size='XS', aes_mode=aes_mode, pipes_in_core=0,
               n_gfmul_ip=1, n_rounds=1, key_pre_exp=False)
    cfg.update(kw)
    return cfg


def _hexfield(cfg_val, n_bytes, what):
    if re.fullmatch(r"^[0-9A-F]+$", cfg_val) is None:
        raise ValueError("%s is not an hexadecimal number" % what)
    # right align, pad with 0s, truncate to the field width (tb/gcm_gctr.py:256,266)
    return '{:0>{width}.{max}}'.format(cfg_val, width=2 * n_bytes, max=2 * n_bytes)


def config_data(cfg):
    """tb/gcm_gctr.py:229-332 -- returns (key dict, iv dict, aad_n_bytes, pt_n_bytes, delays)."""
    if cfg['aes_mode'] == 'ALL':
        cfg['aes_mode'] = random.choice(['128', '192', '256'])
    key = {'n_bytes': {'128': 16, '192': 24}.get(cfg['aes_mode'], 32)}
    iv = {'n_bytes': 12}
    if cfg['iv'] == RANDOM_PARAM:
        cfg['iv'] = ''.join(['{:X}'.format(random.randint(0, 16)) for _ in range(24)])        # sic: 0..16 inclusive
    iv['data'] = _hexfield(cfg['iv'], iv['n_bytes'], 'IV')
    if cfg['key'] == RANDOM_PARAM:
        cfg['key'] = ''.join(['{:X}'.format(random.randint(0, 16)) for _ in range(64)])
    key['data'] = _hexfield(cfg['key'], key['n_bytes'], 'Key')
    n = [int(random.betavariate(.1, .1) * cfg['max_n_byte']) for _ in range(2)]
    for i, name in enumerate(('aad', 'data')):
        if cfg[name] == EMPTY_PARAM:
            n[i] = 0
        elif cfg[name] != RANDOM_PARAM:
            n[i] = (len(cfg[name]) + 1) >> 1
    delays = random.randint(0, 31)
    if cfg['enc_dec'] == 'dec':
        delays &= ~(1 << 2)
    return key, iv, n[0], n[1], delays


def _beats(cfg_val, n_bytes):
    """tb/gcm_gctr.py:340-415 for one stream: list of <=16-byte transactions."""
    out = []
    if cfg_val == RANDOM_PARAM:
        for _ in range(n_bytes >> 4):
            out.append(bytes.fromhex('{:032X}'.format(random.randint(0, (2 ** 128) - 1))))
        rem = n_bytes & 0xF
        if rem:
            out.append(bytes.fromhex('{:0{width}X}'.format(random.randint(0, (2 ** (8 * rem)) - 1), width=2 * rem)))
    elif cfg_val != EMPTY_PARAM:
        for i in range(0, len(cfg_val), 32):
            chunk = cfg_val[i:i + 32]
            if len(chunk) & 1:
                chunk += '0'
            out.append(bytes.fromhex(chunk))
    return out


def replay(cfg, model_cls, dut):
    """Run one test the way tb/gcm_test.py does.  -> dict(ok, n_aad, n_data, tag) ; raises AssertionError on mismatch."""
    cfg = dict(cfg)
    random.seed(cfg['seed'])
    key, iv, aad_n, pt_n, _delays = config_data(cfg)
    model = model_cls(key, iv, cfg['enc_dec'])                                   # tb/gcm_test.py:45
    aad_beats = _beats(cfg['aad'], aad_n)
    data_beats = _beats(cfg['data'], pt_n)
    kb = int(key['data'], 16).to_bytes(key['n_bytes'], 'big')
    ivb = int(iv['data'], 16).to_bytes(iv['n_bytes'], 'big')
    dec = cfg['enc_dec'] != 'enc'
    dut_out, dut_tag = dut(kb, ivb, b"".join(aad_beats), b"".join(data_beats), dec)
    # monitors: one callback per beat (tb/gcm_sequencer.py:129-140,162-173,194-205), then the tag (:231)
    for b in aad_beats:
        model.load_aad(b)
    for b in data_beats:
        (model.load_cipher_text if dec else model.load_plain_text)(b)
    model.get_tag(dut_tag)
    # scoreboard (tb/gcm_test.py:88-94): DUT emissions against the model's expected lists, in order
    off = 0
    for exp in model.data_out:
        assert dut_out[off:off + len(exp)] == exp, "data_out mismatch at byte %d" % off
        off += len(exp)
    assert off == len(dut_out)
    assert model.tag == [dut_tag], "tag mismatch"
    return dict(ok=True, n_aad=len(b"".join(aad_beats)), n_data=off, tag=dut_tag.hex(), aes_mode=cfg['aes_mode'])
