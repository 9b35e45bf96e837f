#!/usr/bin/env python3
"""Extracts FACTS (data, not code) from the reference tree and writes tests/golden/ref_facts.json.

Runs only where /root/reference exists (the build container); the JSON it writes is a committed fixture that the CPU tests
compare this build's constants, JSON member names and Fiat-Shamir transcripts with (tests/test_ref_facts.py).  What it reads,
by regular expression, from /root/reference/src:
  * curves/instances.ts       group names and curve constants (p, a, b | d, order, generator)
  * every `hashPoints('SHA-256', [ ... ])` call with a literal list: file, line, the identifiers in order; and the expressions
    the non-literal calls are fed with (exp.ts `arr`, gk.ts `commitments` / concat chain)
  * every class decorated with @jsonObject: its @jsonMember / @jsonArrayMember names in declaration order
  * curves/group.ts           bytes of the hash that make a challenge; bignum/big.ts serdeBigInt prefix
  * zkpAttestList.ts          default secLevel and the verifier's fixed repetition count
    python tools/ref_constants.py [/root/reference] [out.json]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(ref, rel):
    return open(os.path.join(ref, 'src', rel)).read()


def groups(ref):
    txt = read(ref, 'curves/instances.ts')
    out = {}
    for m in re.finditer(r"export const (\w+) = new (\w+)\(\s*'([\w-]+)',(.*?)\n\)", txt, re.S):
        nums = re.findall(r"BigInt\('(0x[0-9a-fA-F]+)'\)", m.group(4))
        out[m.group(1)] = {'class': m.group(2), 'name': m.group(3), 'constants': [hex(int(n, 16)) for n in nums]}
    return out


def hash_calls(ref):
    calls = []
    for rel in ('commit/equality.ts', 'commit/mult.ts', 'exp/exp.ts', 'proofGK/gk.ts'):
        txt = read(ref, rel)
        for m in re.finditer(r"hashPoints\('SHA-256',\s*(\[[^\]]*\]|[^\n]+?)\)[,\n]", txt):
            arg = m.group(1).strip()
            line = txt.count('\n', 0, m.start()) + 1
            if arg.startswith('['):
                calls.append({'file': rel, 'line': line, 'points': [a.strip() for a in arg[1:-1].split(',') if a.strip()]})
            else:
                calls.append({'file': rel, 'line': line, 'expr': arg})
    # how the non-literal arrays are filled
    exp = read(ref, 'exp/exp.ts')
    fills = re.findall(r"arr\[[^\]]+\] = ([^\n]+)|arr = \[([^\]]*)\]|arr\.push\(([^)]*)\)", exp)
    gk = read(ref, 'proofGK/gk.ts')
    m = re.search(r"commitments = ([^\n,]+(?:\.concat\([^)]*\))+)", gk)
    return calls, {'exp_arr': [next(x for x in f if x) for f in fills], 'gk_commitments': m.group(1) if m else None}


def json_members(ref):
    out = {}
    for rel in ('zkpAttestList.ts', 'exp/exp.ts', 'exp/pointAdd.ts', 'commit/mult.ts', 'commit/equality.ts', 'commit/pedersen.ts', 'proofGK/gk.ts',
                'curves/group.ts', 'curves/weier.ts', 'curves/edwards.ts'):
        txt = read(ref, rel)
        heads = list(re.finditer(r"export (?:abstract )?class (\w+)", txt))
        for k, cm in enumerate(heads):
            body = txt[cm.end():heads[k + 1].start() if k + 1 < len(heads) else len(txt)]
            body = re.split(r"\n\s*constructor\(", body)[0]   # decorated members come before the constructor
            members = []
            for mm in re.finditer(r"@(jsonMember|jsonArrayMember)\((.*?)\)\s*(?:public |readonly |private )*(\w+)[?!]?:", body, re.S):
                opt = 'isRequired: true' not in mm.group(2) and 'serdeBigInt' not in mm.group(2)
                members.append({'name': mm.group(3), 'array': mm.group(1) == 'jsonArrayMember', 'optional': opt})
            if members:
                out[cm.group(1)] = members
    return out


def misc(ref):
    g = read(ref, 'curves/group.ts')
    z = read(ref, 'zkpAttestList.ts')
    b = read(ref, 'bignum/big.ts')
    e = read(ref, 'exp/exp.ts')
    return {
        'challenge_bytes': int(re.search(r"hash\.slice\(0, (\d+)\)", g).group(1)),
        'default_sec_level': int(re.search(r"generateParamsList\(secLevel = (\d+)\)", z).group(1)),
        'verify_reps': int(re.search(r"verifyExp\([^;]*?,\s*(\d+),\s*Q", z, re.S).group(1)),
        'bigint_prefix': re.search(r"let s = '(0x)'", b).group(1),
        'exp_challenge_lsb_first': bool(re.search(r"isOdd\(challenge\)", e) and re.search(r"challenge >>= BigInt\(1\)|challenge = challenge >> BigInt\(1\)|challenge >>= 1n", e)),
        'point_prefix_byte': 4 if "0x04" in read(ref, 'curves/weier.ts') else None,
    }


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'tests', 'golden', 'ref_facts.json')
    calls, fills = hash_calls(ref)
    facts = {'source': 'cloudflare/zkp-ecdsa src/ (extracted by tools/ref_constants.py; data only)', 'groups': groups(ref), 'hash_points_calls': calls,
             'hash_points_arrays': fills, 'json_members': json_members(ref), 'misc': misc(ref)}
    json.dump(facts, open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out, '-', len(facts['groups']), 'groups,', len(calls), 'hashPoints calls,', len(facts['json_members']), 'classes')


if __name__ == '__main__':
    main()
