#!/usr/bin/env python3
"""Reflow a Markdown file to lines of at most WIDTH characters without changing its content: paragraphs and list items are re-wrapped (continuation lines keep the
item's indentation), fenced code blocks and short table rows stay as they are, and a table whose rows exceed WIDTH is rewritten as one bullet per row
("**first cell** - second cell - ...", wrapped) under its header cells, since a Markdown table row cannot be broken.
    python tools/design_reflow.py IN.md OUT.md [WIDTH = 160]"""
import re
import sys
import textwrap


def cells(row):
    out, cur, tick = [], "", False
    for ch in row.strip()[1:]:
        if ch == "`":
            tick = not tick
        if ch == "|" and not tick:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def wrap(text, width, first="", rest=""):
    return textwrap.fill(text, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def reflow(lines, width):
    out, i = [], 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("```"):
            out.append(ln)
            i += 1
            while i < len(lines) and not lines[i].startswith("```"):
                out.append(lines[i]); i += 1
            if i < len(lines):
                out.append(lines[i]); i += 1
            continue
        if ln.lstrip().startswith("|"):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            block = lines[i:j]
            if max(len(b) for b in block) <= width:
                out.extend(block)
            else:
                head = cells(block[0])
                body = [b for b in block[1:] if not re.match(r"^\s*\|[\s:|-]+\|?\s*$", b)]
                out.append(wrap("(table, one bullet per row: " + " | ".join(f"**{h}**" for h in head) + ")", width))
                out.append("")
                for b in body:
                    cs = cells(b)
                    text = " - ".join(([f"**{cs[0]}**"] if cs else []) + [c for c in cs[1:] if c])
                    out.append(wrap(text, width, "* ", "  "))
            i = j
            continue
        if not ln.strip() or ln.startswith("#") or ln.startswith("---"):
            out.append(ln); i += 1
            continue
        # a paragraph or a list item with its continuation lines
        m = re.match(r"^(\s*)([*+-]|\d+\.)\s+", ln)
        indent = (m.group(1) + " " * (len(m.group(2)) + 1)) if m else re.match(r"^\s*", ln).group(0)
        first = m.group(0) if m else indent
        text = ln[len(first):] if m else ln.strip()
        i += 1
        while i < len(lines):
            nx = lines[i]
            if (not nx.strip() or nx.startswith("#") or nx.lstrip().startswith("|") or nx.startswith("```") or re.match(r"^\s*([*+-]|\d+\.)\s+", nx)):
                break
            text += " " + nx.strip()
            i += 1
        out.append(wrap(text, width, first, indent))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 160
    lines = open(src).read().split("\n")
    res = "\n".join(reflow(lines, width)).split("\n")
    open(dst, "w").write("\n".join(res))
    print(src, "->", dst, ":", sum(len(l) > width for l in lines), "long lines before,", sum(len(l) > width for l in res), "after;", len(lines), "->", len(res), "lines")


if __name__ == "__main__":
    main()
