'use strict';
/*
 * strip_ts.js -- removes TypeScript type syntax from the reference's source text IN MEMORY so that node 12 can run the
 * reference's own control flow (the image has no tsc / esbuild / vite).  Build-container cross-check only (see run.js
 * for what this is and is not).  It covers exactly the TypeScript subset the reference's render-path files use:
 *   import / export, interface and type declarations, class member modifiers (public / private / protected / readonly /
 *   abstract / static), `implements`, abstract members, typed and optional parameters, typed variable declarations,
 *   return types (including one object-type literal), typed class properties (with or without a terminating `;`),
 *   generic arguments in `new X<T>()` / `Array<T>`.
 * Anything else makes the generated JavaScript fail to parse, loudly (run.js compiles every module before running).
 */

function tokenize(src) {
  const toks = [];
  let i = 0;
  const n = src.length;
  const isIdStart = (c) => /[A-Za-z_$]/.test(c);
  const isId = (c) => /[A-Za-z0-9_$]/.test(c);
  while (i < n) {
    const c = src[i];
    if (c === '\n') { toks.push({ t: 'nl', v: '\n' }); i++; continue; }
    if (/\s/.test(c)) { let j = i; while (j < n && /[ \t\r\f\v]/.test(src[j])) j++; toks.push({ t: 'ws', v: src.slice(i, j) }); i = j; continue; }
    if (c === '/' && src[i + 1] === '/') { let j = i; while (j < n && src[j] !== '\n') j++; toks.push({ t: 'com', v: src.slice(i, j) }); i = j; continue; }
    if (c === '/' && src[i + 1] === '*') { const j = src.indexOf('*/', i + 2); const e = j < 0 ? n : j + 2; toks.push({ t: 'com', v: src.slice(i, e) }); i = e; continue; }
    if (c === '"' || c === "'" || c === '`') {
      let j = i + 1;
      while (j < n && src[j] !== c) { if (src[j] === '\\') j++; j++; }
      toks.push({ t: 'str', v: src.slice(i, j + 1) }); i = j + 1; continue;
    }
    if (isIdStart(c)) { let j = i; while (j < n && isId(src[j])) j++; toks.push({ t: 'id', v: src.slice(i, j) }); i = j; continue; }
    if (/[0-9]/.test(c) || (c === '.' && /[0-9]/.test(src[i + 1] || ''))) {
      let j = i; while (j < n && /[0-9a-fA-FxXeE._+\-]/.test(src[j]) && !((src[j] === '+' || src[j] === '-') && !/[eE]/.test(src[j - 1]))) j++;
      toks.push({ t: 'num', v: src.slice(i, j) }); i = j; continue;
    }
    const three = src.slice(i, i + 3), two = src.slice(i, i + 2);
    if (['===', '!==', '...', '>>>', '<<=', '>>=', '**='].includes(three)) { toks.push({ t: 'p', v: three }); i += 3; continue; }
    if (['=>', '==', '!=', '<=', '>=', '&&', '||', '++', '--', '+=', '-=', '*=', '/=', '%=', '<<', '>>', '**', '|=', '&=', '^='].includes(two)) { toks.push({ t: 'p', v: two }); i += 2; continue; }
    toks.push({ t: 'p', v: c }); i++;
  }
  return toks;
}

const SKIP = (t) => t.t === 'ws' || t.t === 'nl' || t.t === 'com';

function strip(src, fileName) {
  const T = tokenize(src);
  const out = [];            // tokens kept
  const exportsList = [];
  const N = T.length;
  const next = (i) => { let j = i + 1; while (j < N && SKIP(T[j])) j++; return j; };       // next significant token index
  const prevOut = () => { for (let k = out.length - 1; k >= 0; k--) if (!SKIP(out[k])) return out[k]; return null; };
  const is = (i, v) => i < N && T[i].v === v && T[i].t !== 'str';
  const MODS = new Set(['public', 'private', 'protected', 'readonly', 'abstract', 'override']);

  // skips a type starting at token i; returns the index of the first token that is not part of it.
  // stops (depth 0) at one of the terminator values; `nlStops`: a newline at depth 0 ends the type when the next
  // significant token cannot continue a type (class properties without `;`).
  function skipType(i, stops, nlStops) {
    let depth = 0;
    let j = i;
    let lastSig = null;
    while (j < N) {
      const tk = T[j];
      if (tk.t === 'nl' && nlStops && depth === 0 && lastSig) {
        const k = next(j);
        const nv = k < N ? T[k].v : '';
        if (!['|', '&', '=>', '<', '[', '.', ','].includes(nv) && !['|', '&', '=>', '<', ':', ','].includes(lastSig.v)) return j;
      }
      if (SKIP(tk)) { j++; continue; }
      if (tk.t !== 'str') {
        if (depth === 0 && stops.includes(tk.v) && !(tk.v === '{' && lastSig === null && false)) {
          if (tk.v === '{' && stops.includes('{')) {
            // an object-type literal or the body?  `{...}` followed by `{` (or `=>`) is a type literal
            if (lastSig === null || ['|', '&', ':', '<', ','].includes(lastSig.v)) {
              let d = 0, k = j;
              for (; k < N; k++) { if (T[k].t === 'str') continue; if (T[k].v === '{') d++; else if (T[k].v === '}') { d--; if (d === 0) break; } }
              const a = next(k);
              if (a < N && (T[a].v === '{' || T[a].v === '=>' || T[a].v === '|')) { j = k + 1; lastSig = T[k]; continue; }
            }
          }
          return j;
        }
        if (tk.v === '=>' && depth === 0 && !stops.includes('=>')) { lastSig = tk; j++; continue; }
        if ('<([{'.includes(tk.v) && tk.v.length === 1) depth++;
        else if ('>)]}'.includes(tk.v) && tk.v.length === 1) { if (depth === 0) return j; depth--; }
        else if (tk.v === '>>') { depth -= 2; if (depth < 0) return j; }
      }
      lastSig = tk;
      j++;
    }
    return j;
  }

  // copies a parameter list starting at `(` (index i), dropping `?`, `: Type` and modifiers; returns index after `)`
  function copyParams(i) {
    out.push(T[i]);
    let j = i + 1, depth = 0;
    let expectName = true;
    while (j < N) {
      const tk = T[j];
      if (SKIP(tk)) { out.push(tk); j++; continue; }
      if (tk.t !== 'str') {
        if (tk.v === ')' && depth === 0) { out.push(tk); return j + 1; }
        if (expectName && depth === 0 && tk.t === 'id' && MODS.has(tk.v)) { j++; continue; }
        if (depth === 0 && tk.v === '?' && is(next(j), ':')) { j++; continue; }   // optional parameter
        if (depth === 0 && tk.v === '?' && (is(next(j), ',') || is(next(j), ')'))) { j++; continue; }
        if (depth === 0 && tk.v === ':') { j = skipType(j + 1, ['=', ',', ')'], false); expectName = false; continue; }
        if (depth === 0 && tk.v === ',') { expectName = true; out.push(tk); j++; continue; }
        if ('([{'.includes(tk.v) && tk.v.length === 1) depth++;
        else if (')]}'.includes(tk.v) && tk.v.length === 1) depth--;
        if (depth === 0 && tk.v === '=') expectName = false;
      }
      out.push(tk);
      j++;
    }
    return j;
  }

  // after a parameter list: optional return type
  function skipReturnType(j) {
    const k = next(j - 1);
    if (is(k, ':')) return skipType(k + 1, ['{', '=>', ';'], false);
    return j;
  }

  // is the `(` at index i the start of an arrow function's parameter list?
  function arrowParamsAt(i) {
    let d = 0, k = i;
    for (; k < N; k++) { if (T[k].t === 'str') continue; if (T[k].v === '(') d++; else if (T[k].v === ')') { d--; if (d === 0) break; } }
    const a = next(k);
    if (is(a, '=>')) return true;
    if (is(a, ':')) { const e = skipType(a + 1, ['=>', ';', '{', ',', ')'], false); return is(e, '=>'); }
    return false;
  }

  const classDepth = [];  // brace depths at which a class body is open
  let brace = 0;
  let i = 0;
  let pendingClass = false;
  while (i < N) {
    const tk = T[i];
    if (SKIP(tk) || tk.t === 'str' || tk.t === 'num') { out.push(tk); i++; continue; }
    const inClassBody = classDepth.length && classDepth[classDepth.length - 1] === brace;
    const p = prevOut();
    const atStmtStart = !p || [';', '{', '}'].includes(p.v) || /[;}]$/.test(p.v) || (p.t === 'id' && ['else'].includes(p.v));

    // ---- imports
    if (tk.t === 'id' && tk.v === 'import' && atStmtStart) {
      let j = next(i);
      if (T[j].t === 'str') { out.push({ t: 'id', v: 'require_(' + T[j].v + ');' }); j = next(j); if (is(j, ';')) j++; i = j; continue; }
      let spec = '';
      while (!(T[j].t === 'id' && T[j].v === 'from')) { spec += T[j].v; j++; }
      j = next(j);
      const mod = T[j].v;
      j = next(j);
      if (is(j, ';')) j++;
      spec = spec.trim();
      if (spec.startsWith('{')) spec = spec.replace(/\s+as\s+/g, ': ').replace(/\btype\s+/g, '');
      else if (spec.startsWith('*')) spec = spec.replace(/^\*\s*as\s+/, '');
      else spec = '{ default: ' + spec + ' }';
      out.push({ t: 'id', v: 'const ' + spec + ' = require_(' + mod + ');' });
      i = j; continue;
    }
    // ---- export
    if (tk.t === 'id' && tk.v === 'export' && atStmtStart) {
      let j = next(i);
      if (T[j].t === 'id' && T[j].v === 'default') throw new Error(fileName + ': export default is not handled');
      let k = j;
      if (T[k].t === 'id' && T[k].v === 'abstract') k = next(k);
      if (T[k].t === 'id' && ['class', 'function', 'const', 'let', 'var'].includes(T[k].v)) exportsList.push(T[next(k)].v);
      i = j; continue;  // the declaration itself is handled by the main loop
    }
    // ---- interface / type alias
    if (tk.t === 'id' && tk.v === 'interface' && atStmtStart && T[next(i)].t === 'id') {
      let j = i; while (!is(j, '{')) j++;
      let d = 0; for (; j < N; j++) { if (T[j].t === 'str') continue; if (T[j].v === '{') d++; else if (T[j].v === '}') { d--; if (d === 0) break; } }
      i = j + 1; continue;
    }
    if (tk.t === 'id' && tk.v === 'type' && atStmtStart && T[next(i)].t === 'id' && is(next(next(i)), '=')) {
      let j = skipType(next(next(i)) + 1, [';'], true);
      if (is(j, ';')) j++;
      i = j; continue;
    }
    // ---- class header
    if (tk.t === 'id' && tk.v === 'abstract' && T[next(i)].v === 'class') { i = next(i); continue; }
    if (tk.t === 'id' && tk.v === 'class') { pendingClass = true; out.push(tk); i++; continue; }
    if (pendingClass && tk.t === 'id' && tk.v === 'implements') { let j = i; while (!is(j, '{')) j++; i = j; continue; }
    if (tk.v === '{') {
      brace++;
      if (pendingClass) { classDepth.push(brace); pendingClass = false; }
      out.push(tk); i++; continue;
    }
    if (tk.v === '}') {
      if (classDepth.length && classDepth[classDepth.length - 1] === brace) classDepth.pop();
      brace--; out.push(tk); i++; continue;
    }
    // ---- class members
    if (inClassBody && tk.t === 'id' && atStmtStart) {
      let j = i, isAbstract = false, isStatic = false;
      while (T[j].t === 'id' && (MODS.has(T[j].v) || T[j].v === 'static') && !is(next(j), '(') && !is(next(j), '=') && !is(next(j), ':') && !is(next(j), ';')) {
        if (T[j].v === 'abstract') isAbstract = true;
        if (T[j].v === 'static') isStatic = true;
        j = next(j);
      }
      if (isAbstract) {  // abstract member: remove up to and including `;`
        let d = 0;
        for (; j < N; j++) { if (T[j].t === 'str') continue; if ('([{'.includes(T[j].v)) d++; else if (')]}'.includes(T[j].v)) d--; else if (T[j].v === ';' && d === 0) break; }
        i = j + 1; continue;
      }
      if (isStatic) out.push({ t: 'id', v: 'static' }, { t: 'ws', v: ' ' });
      // member name (identifier, `get`/`set` accessor, or `async`)
      let nameIdx = j;
      if (T[nameIdx].t === 'id' && ['get', 'set', 'async'].includes(T[nameIdx].v) && T[next(nameIdx)].t === 'id' && !is(next(nameIdx), '(')) {
        out.push(T[nameIdx], { t: 'ws', v: ' ' }); nameIdx = next(nameIdx);
      }
      let a = next(nameIdx);
      if (is(a, '(')) {  // method
        out.push(T[nameIdx]);
        let e = copyParams(a);
        e = skipReturnType(e);
        i = e; continue;
      }
      // property: name [?|!] [: Type] [= init] [;]
      out.push(T[nameIdx]);
      if (is(a, '?') || is(a, '!')) a = next(a);
      if (is(a, ':')) a = skipType(a + 1, ['=', ';'], true);
      if (!is(a, '=') && !is(a, ';')) out.push({ t: 'p', v: ';' });  // a property the source ends with a line break only
      i = a; continue;
    }
    // ---- functions
    if (tk.t === 'id' && tk.v === 'function') {
      out.push(tk);
      let j = i + 1;
      while (!is(j, '(')) { out.push(T[j]); j++; }
      let e = copyParams(j);
      e = skipReturnType(e);
      i = e; continue;
    }
    if (tk.v === '(' && arrowParamsAt(i)) {
      let e = copyParams(i);
      e = skipReturnType(e);
      i = e; continue;
    }
    // ---- variable declarations: const x: T = ...
    if (tk.t === 'id' && ['const', 'let', 'var'].includes(tk.v)) {
      out.push(tk);
      let j = next(i);
      for (let k = i + 1; k < j; k++) out.push(T[k]);
      if (T[j].t === 'id') {
        out.push(T[j]);
        const a = next(j);
        if (is(a, ':')) { i = skipType(a + 1, ['=', ';', ',', ')'], true); if (T[i] && T[i].t === 'id' && ['of', 'in'].includes(T[i].v)) { /* for-of */ } continue; }
        i = j + 1; continue;
      }
      i = j; continue;
    }
    // ---- generic arguments after an identifier: new Set<Primitive>() / Array<number>(n)
    if (tk.v === '<' && p && p.t === 'id' && /^[A-Z]/.test(p.v)) {
      let d = 0, k = i, ok = true;
      for (; k < N; k++) {
        if (T[k].v === '<') d++; else if (T[k].v === '>') { d--; if (d === 0) break; }
        else if (!(SKIP(T[k]) || T[k].t === 'id' || [',', '[', ']', '|', '.'].includes(T[k].v))) { ok = false; break; }
      }
      if (ok && k < N && (is(next(k), '(') || is(next(k), ';') || is(next(k), ')') || is(next(k), '='))) { i = k + 1; continue; }
    }
    // ---- non-null assertion: postfix `!` after an expression, before something that cannot follow a prefix operator
    if (tk.v === '!' && p && ([')', ']'].includes(p.v) || (p.t === 'id' && !['return', 'typeof', 'case', 'in', 'of', 'void', 'delete', 'new', 'throw', 'else'].includes(p.v))) &&
        [';', ')', ',', '.', '[', ']', '}'].includes(T[next(i)].v)) { i++; continue; }
    out.push(tk);
    i++;
  }
  let js = out.map((t) => t.v).join('');
  js += '\n' + exportsList.map((nm) => 'exports_.' + nm + ' = ' + nm + ';').join('\n') + '\n';
  return js;
}

module.exports = { strip };
