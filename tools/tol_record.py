"""WF_TOL_LOG of a whole GPU-suite run -> tests/golden/tolerances_mi355x.json: the error every (name, pytest case) pair measured on an MI355X
(the max, should a case log a name more than once).  tests/_tol.within then holds every case to 2 x its own record.
    python tools/tol_record.py gpurun_out/final/tolerances.txt [--merge]"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "tolerances_mi355x.json")
table = {}
if "--merge" in sys.argv and os.path.exists(OUT):
    table = json.load(open(OUT))
new = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(.*) measured (\S+) bound (\S+) ratio ", line)
    if not m or " | " not in m.group(1):
        continue
    k, v = m.group(1), float(m.group(2))
    new[k] = max(new.get(k, 0.0), v)
table.update(new)
json.dump(dict(sorted(table.items())), open(OUT, "w"), indent=0)
print(f"{len(new)} cases recorded ({len(table)} in the table) -> {OUT}")
