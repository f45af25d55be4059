"""Re-wraps a markdown file at 120 columns (paragraphs of 300-1 500 characters per line are unreadable in a diff): paragraphs
and bullets are wrapped with a hanging indent, fenced code and short table rows are left alone, table rows longer than the limit become
bullets ("* **first cell** -- the other cells").  usage: python tools/wrap_md.py IN [OUT]"""
import sys
import textwrap

W = 120


def wrap(text, first, rest):
    return textwrap.fill(" ".join(text.split()), width=W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def main(src, dst):
    out, para, fence = [], [], False

    def flush():
        if para:
            out.append(wrap(" ".join(para), "", ""))
            para.clear()
    lines = open(src).read().split("\n")
    table_long = False
    for ln in lines:
        s = ln.rstrip()
        if s.lstrip().startswith("```"):
            flush(); fence = not fence; out.append(s); continue
        if fence:
            out.append(s); continue
        st = s.lstrip()
        if not st:
            flush(); out.append(""); continue
        if st.startswith("|"):
            flush()
            cells = [c.strip() for c in st.strip("|").split("|")]
            if all(set(c) <= set("-: ") for c in cells):
                if not table_long:
                    out.append(s)
                continue
            if len(s) <= W:
                table_long = False
                out.append(s); continue
            # a long row: as a bullet (its header row, if short, has been emitted already: drop the separator that followed)
            if out and out[-1].startswith("|") and set(out[-1].replace("|", "").strip()) <= set("-: "):
                out.pop()
            table_long = True
            out.append(wrap("**%s** -- %s" % (cells[0], " | ".join(c for c in cells[1:] if c)), "* ", "  "))
            continue
        table_long = False
        if st.startswith("#"):
            flush(); out.append(s); continue
        ind = len(s) - len(st)
        if st[:2] in ("* ", "- ") or (st[:1].isdigit() and st[1:3] in (". ", ") ")):
            flush()
            mark = st[:2] if st[:2] in ("* ", "- ") else st[:3]
            out.append(wrap(st[len(mark):], " " * ind + mark, " " * (ind + len(mark))))
            continue
        if ind >= 2 and out and out[-1].startswith((" " * (ind - 2) + "* ", " " * (ind - 2) + "- ", " " * ind)) and not para:
            # continuation line of a bullet: join onto it
            last = out.pop()
            li = len(last) - len(last.lstrip())
            mark = last.lstrip()[:2] if last.lstrip()[:2] in ("* ", "- ") else ""
            body = last.lstrip()[len(mark):] if mark else last.lstrip()
            if mark:
                out.append(wrap(body + " " + st, " " * li + mark, " " * (li + 2)))
            else:
                out.append(wrap(body + " " + st, " " * li, " " * li))
            continue
        para.append(st)
    flush()
    open(dst, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
