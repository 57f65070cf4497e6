// Node 12 ESM loader hook: treat every .js file under the reference tree as an
// ES module, so the reference's lib/*.js and src/*.js can be imported in place
// (read-only) without copying them anywhere.
export async function getFormat(url, context, defaultGetFormat) {
  if (url.startsWith('file:///root/reference/') && url.endsWith('.js')) {
    return { format: 'module' }
  }
  return defaultGetFormat(url, context, defaultGetFormat)
}
