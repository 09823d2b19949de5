#!/usr/bin/env python3
"""Regenerate dynearthsol_amd/csrc/host/cfg_options.inc from the reference's option
declarations (input.cxx:36-905).  Only the user-facing key names, value types and
default values are extracted -- they are the .cfg interface a drop-in must accept.
Dev-time tool: needs /root/reference, never runs on the GPU box."""
import re, sys, os
ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
src = open(os.path.join(ref, 'input.cxx')).read()
body = src[src.index('cfg.add_options()'):src.index('static void read_parameters_from_file')]
body = '\n'.join(l for l in body.split('\n') if not l.strip().startswith('//'))
pat = re.compile(r'\(\s*"([a-zA-Z_0-9\.]+)"\s*,\s*po::value<([^>]+)>\(([^)]*)\)'
                 r'((?:->[a-z_]+\((?:[^()]|\([^()]*\))*\))*)', re.S)
tmap = {'std::string': 'STR', 'int': 'INT', 'double': 'DBL', 'bool': 'BOOL', 'uint': 'UINT'}
out = []
for m in pat.finditer(body):
    name, typ, _tgt, mods = m.groups()
    mm = re.search(r'default_value\(((?:[^()]|\([^()]*\))*)\)', mods)
    d = mm.group(1).strip().strip('"') if mm else None
    if d == 'std::numeric_limits<double>::max()':
        d = '1.7976931348623157e308'
    ds = 'nullptr' if d is None else '"%s"' % d
    out.append('    {"%s", CFG_%s, %s, %s},' % (name, tmap[typ.strip()], ds,
                                               'true' if 'required' in mods else 'false'))
here = os.path.dirname(os.path.abspath(__file__))
dst = os.path.join(here, '..', 'dynearthsol_amd', 'csrc', 'host', 'cfg_options.inc')
open(dst, 'w').write(
    "// Option table of the .cfg front-end: key, value type, default (nullptr = none), required.\n"
    "// Keys, types and defaults are the reference's user-facing interface (input.cxx:36-905);\n"
    "// regenerate with tools/gen_cfg_options.py when the reference adds options.\n"
    + '\n'.join(out) + '\n')
print('%d options -> %s' % (len(out), dst))
