"""The reference test flow replayed without a simulator (tests/replay_harness.py).
CPU: the harness itself, with a tiny oracle-backed model class.  GPU: the drop-in gcm_model class."""
import pytest

import replay_harness as rh


def oracle_dut(orc):
    def dut(key, iv, aad, data, dec):
        f = orc.Fast(key)
        return f.decrypt(iv, aad, data) if dec else f.encrypt(iv, aad, data)
    return dut


class _OracleModel:
    """gcm_model-shaped class over the oracle: lets the harness be tested on a box without a GPU."""
    def __init__(self, key, icb, ed):
        from oracle import oracle as O
        self.ed, self.data_out, self.tag = ed, [], []
        self.f = O.Fast(int(key['data'], 16).to_bytes(key['n_bytes'], 'big'))
        self.f.begin(int(icb['data'], 16).to_bytes(icb['n_bytes'], 'big'), dec=(ed != 'enc'))

    def load_aad(self, aad): self.f.aad(aad)
    def load_plain_text(self, pt): self.data_out.append(bytes(self.f.update(pt)))
    def load_cipher_text(self, ct): self.data_out.append(bytes(self.f.update(ct)))

    def get_tag(self, tag):
        mine = self.f.final()
        self.tag.append(mine if self.ed == 'enc' or mine == tag else bytes(b ^ 0xFF for b in tag))


def test_harness_draws_and_flow_cpu(orc):
    seen = set()
    for seed in range(1, 25):
        for mode, ed in (('128', 'enc'), ('192', 'dec'), ('256', 'enc')):
            r = rh.replay(rh.default_config(seed * 7919 + int(mode), aes_mode=mode, enc_dec=ed), _OracleModel, oracle_dut(orc))
            assert r["ok"]
            seen.add((r["n_aad"] == 0, r["n_data"] == 0))
    assert len(seen) >= 3                       # the U-shaped length law hits the empty corners


def test_directed_readme_vectors_cpu(orc):
    cfg = rh.default_config(1, aes_mode='128', key='AD7A2BD03EAC835A6F620FDCB506B345', iv='12153524C0895E81B2C28465',
                            aad='D609B1F056637A0D46DF998D88E52E00B2C2846512153524C0895E81',
                            data='08000F101112131415161718191A1B1C1D1E1F202122232425262728292A2B2C2D2E2F303132333435363738393A0002')
    assert rh.replay(cfg, _OracleModel, oracle_dut(orc))["tag"].upper() == "4F8D55E7D3F06FD5A13C0C29B9D5B880"
    cfg = rh.default_config(2, aes_mode='256', key='691D3EE909D7F54167FD1CA0B5D769081F2BDE1AEE655FDBAB80BD5295AE6BE7',
                            iv='F0761E8DCD3D000176D457ED', data='EMPTY',
                            aad='E20106D7CD0DF0761E8DCD3D88E5400076D457ED08000F101112131415161718191A1B1C1D1E1F202122232425262728292A2B2C2D2E2F303132333435363738393A0003')
    assert rh.replay(cfg, _OracleModel, oracle_dut(orc))["tag"].upper() == "35217C774BBC31B63166BCF9D4ABED07"


@pytest.mark.gpu
def test_gpu_dropin_under_reference_stimulus(hip, orc):
    from aesgcm_amd import gcm_model
    n = 0
    for seed in range(100, 112):
        for mode in ('128', '192', '256'):
            for ed in ('enc', 'dec'):
                r = rh.replay(rh.default_config(seed * 31 + int(mode), aes_mode=mode, enc_dec=ed), gcm_model.gcm, oracle_dut(orc))
                assert r["ok"]
                n += 1
    assert n == 72


@pytest.mark.gpu
def test_gpu_dropin_medium_size_and_tamper(hip, orc):
    from aesgcm_amd import gcm_model
    r = rh.replay(rh.default_config(424242, aes_mode='256', enc_dec='enc', test_size='medium'), gcm_model.gcm, oracle_dut(orc))
    assert r["ok"]
    # a DUT that returns a wrong tag in dec mode: the model must append the inverted tag (tb/gcm_model.py:49-51)
    def bad_dut(key, iv, aad, data, dec):
        out, tag = oracle_dut(orc)(key, iv, aad, data, dec)
        return out, bytes([tag[0] ^ 1]) + tag[1:]
    with pytest.raises(AssertionError, match="tag mismatch"):
        rh.replay(rh.default_config(99, aes_mode='128', enc_dec='dec'), gcm_model.gcm, bad_dut)
