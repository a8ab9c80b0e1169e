#!/usr/bin/env python3
"""Partial preprocessor used once, in round 5, to delete the experiment switches whose verdict in MEASUREMENTS.md is
"not adopted / slower / +-0" TOGETHER WITH their code paths (kept as the record of how the clean-up was done; the check that
it changed nothing is tools/codeobj_digest.py on the built library before and after).

usage: tools/prune_switches.py FILE... -- NAME=VALUE ... NAME! ...
  NAME=VALUE  the switch is fixed at VALUE: `#ifndef NAME / #define NAME ... / #endif` default blocks go, conditionals on it are
              resolved, remaining uses of NAME in code are replaced by VALUE
  NAME!       the switch is never defined: `#ifdef NAME` branches go, `#ifndef NAME` bodies stay
Conditions that also mention other macros are simplified where that is safe (x && 1, 0 || x ...) and otherwise reported."""
import re
import sys


def simplify(expr, known):
    """-> True / False when decided, or a rewritten expression string."""
    e = expr
    for name, val in known.items():
        e = re.sub(r"defined\s*\(\s*%s\s*\)|defined\s+%s\b" % (name, name), "1" if val is not None else "0", e)
    for name, val in known.items():
        e = re.sub(r"\b%s\b" % name, str(val) if val is not None else "0", e)
    e = e.strip()
    if not re.search(r"[A-Za-z_]\w*", e):
        py = e.replace("&&", " and ").replace("||", " or ")
        py = re.sub(r"!(?!=)", " not ", py)
        return bool(eval(py))
    # partially known: top-level && / || only (no mixed nesting handled beyond parentheses around single terms)
    changed = True
    while changed:
        changed = False
        for pat, rep in ((r"!\s*0\b", "1"), (r"!\s*1\b", "0"), (r"\(\s*([01])\s*\)", r"\1"),
                         (r"\b1\s*&&\s*", ""), (r"\s*&&\s*1\b", ""), (r"\b0\s*\|\|\s*", ""), (r"\s*\|\|\s*0\b", "")):
            n = re.sub(pat, rep, e)
            if n != e:
                e, changed = n, True
    if re.fullmatch(r"[01]", e.strip()):
        return e.strip() == "1"
    if "||" not in e and re.search(r"(^|&&)\s*0\s*($|&&)", e):
        return False
    if "&&" not in e and re.search(r"(^|\|\|)\s*1\s*($|\|\|)", e):
        return True
    return e


def prune(text, known):
    lines = text.split("\n")
    out = []
    # stack entries: dict(kind = "resolved" | "kept", taken = bool (a branch already emitted / chosen), active = bool (emit lines), parent_active)
    stack = []
    active = True
    report = []
    i = 0
    while i < len(lines):
        line = lines[i]
        # join continuation lines of directives
        m = re.match(r"^\s*#\s*(ifdef|ifndef|if|elif|else|endif|define|undef)\b(.*)$", line)
        if not m:
            if active:
                out.append(line)
            i += 1
            continue
        kw, rest = m.group(1), m.group(2)
        code = rest.split("//")[0].strip()
        comment = rest[len(rest.split("//")[0]):] if "//" in rest else ""
        if kw in ("ifdef", "ifndef", "if"):
            if kw == "ifdef":
                cond = "defined(%s)" % code
            elif kw == "ifndef":
                cond = "!defined(%s)" % code
            else:
                cond = code
            r = simplify(cond, known) if active else False
            # default block: #ifndef X / #define X ... / #endif with X known -> drop whole block
            if active and kw == "ifndef" and code in known and known[code] is not None:
                j = i + 1
                while j < len(lines) and not re.match(r"^\s*#\s*endif", lines[j]):
                    j += 1
                body = [l for l in lines[i + 1:j] if l.strip()]
                if all(re.match(r"^\s*#\s*define\s+%s\b" % code, l) or l.strip().startswith("//") for l in body):
                    i = j + 1
                    continue
            if isinstance(r, bool):
                stack.append(dict(kind="resolved", taken=r, active=r and active, parent=active))
                active = r and active
            else:
                if active:
                    new = "#if " + r
                    if kw in ("ifdef", "ifndef") and r == cond:
                        new = line.strip().split("//")[0].rstrip()
                    out.append(new + ("   " + comment if comment and r == cond else ("   " + comment if comment else "")))
                    if r != cond:
                        report.append("line %d: condition rewritten: %s -> %s" % (i + 1, cond, r))
                stack.append(dict(kind="kept", taken=False, active=active, parent=active))
        elif kw == "elif":
            top = stack[-1]
            if top["kind"] == "resolved":
                if top["taken"]:
                    active = False
                    top["active"] = False
                else:
                    r = simplify(code, known) if top["parent"] else False
                    if isinstance(r, bool):
                        top["taken"] = r
                        active = r and top["parent"]
                        top["active"] = active
                    else:
                        # becomes the head of a kept conditional
                        if top["parent"]:
                            out.append("#if " + r + ("   " + comment if comment else ""))
                        top["kind"] = "kept"
                        active = top["parent"]
                        top["active"] = active
            else:
                r = simplify(code, known) if top["parent"] else code
                if top["parent"]:
                    if r is True:
                        out.append("#else" + ("   " + comment if comment else ""))
                        top["kind"] = "kept_else_done"
                    elif r is False:
                        # drop this branch: emit nothing until next elif/else/endif
                        top["skip"] = True
                        active = False
                        i += 1
                        continue
                    else:
                        out.append("#elif " + r + ("   " + comment if comment else ""))
                top["skip"] = False
                active = top["parent"]
        elif kw == "else":
            top = stack[-1]
            if top["kind"] == "resolved":
                active = (not top["taken"]) and top["parent"]
                top["taken"] = True
                top["active"] = active
            elif top["kind"] == "kept_else_done":
                active = False
            else:
                if top["parent"]:
                    out.append(line)
                active = top["parent"]
        elif kw == "endif":
            top = stack.pop()
            if top["kind"] != "resolved" and top["parent"]:
                out.append(line)
            active = top["parent"]
        elif kw in ("define", "undef"):
            name = re.match(r"\s*(\w+)", code).group(1)
            if active and not (name in known):
                out.append(line)
            elif active and name in known:
                report.append("line %d: dropped '#%s %s'" % (i + 1, kw, name))
        i += 1
    text = "\n".join(out)
    for name, val in known.items():
        if val is not None:
            text = re.sub(r"\b%s\b" % name, str(val), text)
    return text, report


if __name__ == "__main__":
    args = sys.argv[1:]
    k = args.index("--")
    files, specs = args[:k], args[k + 1:]
    known = {}
    for s in specs:
        if s.endswith("!"):
            known[s[:-1]] = None
        else:
            n, v = s.split("=", 1)
            known[n] = v
    for f in files:
        src = open(f).read()
        new, report = prune(src, known)
        if new != src:
            open(f, "w").write(new)
        print("%s: %d -> %d lines" % (f, src.count("\n") + 1, new.count("\n") + 1))
        for r in report:
            print("   ", r)
