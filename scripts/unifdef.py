"""Development aid: remove closed development switches from the kernel sources.

    python scripts/unifdef.py --undef RPSF_DEV_WIDE,RPSF_DEV_SPLIT,... file ...   (in place)

Every macro named in --undef is taken as NOT defined (value 0); conditionals whose outcome follows from that alone are resolved
(directive lines and dead branches removed), everything else - including conditionals on macros that are not named - is left as
it is.  Handles #if / #ifdef / #ifndef / #elif / #else / #endif with nesting, `defined(X)`, `!`, `&&`, `||`, comparisons and `&`.
"""
import argparse
import pathlib
import re
import sys

DIRECTIVE = re.compile(r"^\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)$")


def evaluate(expr: str, undef: set[str]):
    """True / False if the expression is decided by the undefined macros alone, None otherwise."""
    expr = re.sub(r"//.*$", "", expr)
    expr = re.sub(r"/\*.*?\*/", "", expr).strip()
    unknown = False

    def repl_defined(m):
        nonlocal unknown
        name = m.group(1) or m.group(2)
        if name in undef:
            return " 0 "
        unknown = True
        return " 0 "

    text = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", repl_defined, expr)

    def repl_ident(m):
        nonlocal unknown
        name = m.group(0)
        if name in undef:
            return "0"
        unknown = True
        return "0"

    text = re.sub(r"\b[A-Za-z_]\w*\b", repl_ident, text)
    if unknown:
        # an unknown operand may still be dominated: evaluate with it False and with it True
        results = set()
        for guess in ("0", "1"):
            def rd(m, g=guess):
                name = m.group(1) or m.group(2)
                return " 0 " if name in undef else f" {g} "
            t = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", rd, expr)
            t = re.sub(r"\b[A-Za-z_]\w*\b", lambda m, g=guess: "0" if m.group(0) in undef else g, t)
            results.add(bool(eval_c(t)))
        return results.pop() if len(results) == 1 else None
    return bool(eval_c(text))


def eval_c(text: str):
    text = text.replace("&&", " and ").replace("||", " or ")
    text = re.sub(r"!(?!=)", " not ", text)
    return eval(text, {"__builtins__": {}}, {})  # noqa: S307 - integers and operators only


def parse(lines, pos, undef):
    """Parse until the matching #elif/#else/#endif of the enclosing group; returns (output lines, next position, terminator)."""
    out = []
    while pos < len(lines):
        line = lines[pos]
        m = DIRECTIVE.match(line)
        if not m:
            out.append(line)
            pos += 1
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("elif", "else", "endif"):
            return out, pos, (kind, rest, line)
        # a conditional group starts here
        if kind == "ifdef":
            cond_text = f"defined({rest.split()[0]})"
        elif kind == "ifndef":
            cond_text = f"!defined({rest.split()[0]})"
        else:
            cond_text = rest
        branches = []  # (value, original directive line, body)
        first_line = line
        value = evaluate(cond_text, undef)
        body, pos, term = parse(lines, pos + 1, undef)
        branches.append((value, first_line, cond_text, body))
        else_body = None
        while term[0] != "endif":
            if term[0] == "elif":
                value = evaluate(term[1], undef)
                line_e, text_e = term[2], term[1]
                body, pos, term = parse(lines, pos + 1, undef)
                branches.append((value, line_e, text_e, body))
            else:  # else
                else_body, pos, term = parse(lines, pos + 1, undef)
        endif_line = term[2]
        pos += 1
        if all(v is None for v, *_ in branches):  # nothing decided: keep the group as written (bodies are already processed)
            for i, (v, dl, ct, b) in enumerate(branches):
                out.append(dl)
                out.extend(b)
            if else_body is not None:
                out.append(re.sub(r"#\s*\w+.*", "#else", branches[0][1].split("#")[0] + "#else"))
                out.extend(else_body)
            out.append(endif_line)
            continue
        kept = []
        final = else_body
        for v, dl, ct, b in branches:
            if v is False:
                continue
            if v is True:
                final = b
                break
            kept.append((ct, b))
        if not kept:
            out.extend(final or [])
            continue
        indent = re.match(r"^\s*", branches[0][1]).group(0)
        for i, (ct, b) in enumerate(kept):
            out.append(f"{indent}#{'if' if i == 0 else 'elif'} {ct.strip()}")
            out.extend(b)
        if final is not None:
            out.append(f"{indent}#else")
            out.extend(final)
        out.append(endif_line)
    return out, pos, ("eof", "", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--undef", required=True)
    ap.add_argument("files", nargs="+")
    a = ap.parse_args()
    undef = set(a.undef.split(","))
    for name in a.files:
        path = pathlib.Path(name)
        lines = path.read_text().split("\n")
        out, pos, term = parse(lines, 0, undef)
        if term[0] != "eof":
            sys.exit(f"{name}: unbalanced conditional at line {pos + 1}")
        path.write_text("\n".join(out))
        print(f"{name}: {len(lines)} -> {len(out)} lines")


if __name__ == "__main__":
    main()
