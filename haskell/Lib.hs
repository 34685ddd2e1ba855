{-# LANGUAGE ForeignFunctionInterface #-}
-- Lib.hs -- drop-in replacement for the reference's src/Lib.hs whose `traces`
-- is computed by libpt.so (HIP kernels on an MI355X) instead of the recursive
-- Haskell bounce loop.  app/Write.hs and app/Web.hs import `traces` unchanged.
--
-- NOT COMPILED in the build image (no GHC there, SURVEY.md F1): this is the
-- binding a maintainer adds; include/pt.h is authoritative for the C side and
-- INTEGRATION.md explains the cabal changes (extra-libraries: pt).
--
-- `bounceRay`, `reflectVector` and `specular` stay exported from the pure
-- module LibPure (the reference's old Lib.hs, renamed) for test/Spec.hs.
module Lib
  ( bounceRay
  , reflectVector
  , specular
  , traces
  ) where

import           Control.Concurrent.MVar (MVar, newMVar, withMVar)
import           Control.Monad          (when)
import           Data.IORef             (IORef, newIORef, readIORef, writeIORef)
import           Data.Array.Repa        ((:.) (..), Array, D, DIM3, Z (..))
import qualified Data.Array.Repa        as R
import qualified Data.Vector            as V
import           Data.Int               (Int32, Int64)
import           Data.Word              (Word32)
import           Foreign
import           Foreign.C.String       (CString, peekCString)
import           Foreign.C.Types
import           System.IO.Unsafe       (unsafeDupablePerformIO, unsafePerformIO)

import           LibPure                (bounceRay, reflectVector, specular)
import           Object                 (Color (..), Form (..), Object (..), Point (..), Vector (..), objects)
import qualified Params
import           Triple                 (Triple (..))

data PtCtx

-- pt_params / pt_object, field for field as in include/pt.h
data PtParams = PtParams
  { ppHeight, ppWidth :: Int32, ppCameraDepth :: Double, ppMaxBounces :: Int32
  , ppSeed :: Word32, ppDevice, ppSampleOffset, ppSampleStride, ppFlags :: Int32 }

instance Storable PtParams where
  sizeOf _ = 40
  alignment _ = 8
  peek _ = error "PtParams: write-only"
  poke p (PtParams h w d b s dev off str fl) = do
    pokeByteOff p 0 h; pokeByteOff p 4 w; pokeByteOff p 8 d; pokeByteOff p 16 b
    pokeByteOff p 20 s; pokeByteOff p 24 dev; pokeByteOff p 28 off; pokeByteOff p 32 str
    pokeByteOff p 36 fl

newtype PtObject = PtObject Object

instance Storable PtObject where
  sizeOf _ = 112
  alignment _ = 8
  peek _ = error "PtObject: write-only"
  poke p (PtObject (Object (Color (Triple r g b)) emit refl form _name)) = do
    let (kind, Point (Triple px py pz), Vector (Triple nx ny nz), radius) = case form of
          InfinitePlane pt n -> (0 :: Int32, pt, n, 0)
          Disk c n rad       -> (1 :: Int32, c, n, rad)
    pokeByteOff p 0 kind
    pokeByteOff p 4 (if refl then 1 else 0 :: Int32)
    mapM_ (\(k, x) -> pokeByteOff p (8 + 8 * k) (x :: Double))  (zip [0 ..] [px, py, pz])
    mapM_ (\(k, x) -> pokeByteOff p (32 + 8 * k) (x :: Double)) (zip [0 ..] [nx, ny, nz])
    pokeByteOff p 56 (radius :: Double)
    mapM_ (\(k, x) -> pokeByteOff p (64 + 8 * k) (x :: Double)) (zip [0 ..] [r, g, b])
    pokeByteOff p 88 (emit :: Double)
    pokeByteOff p 96 (0 :: Double); pokeByteOff p 104 (0 :: Double)   -- extent: PT_RECTANGLE only (Object.hs:57-61 is commented out)

-- `safe`: a call lasts milliseconds to seconds and `web` runs -threaded
-- (pathtracer.cabal:37); Warp's other threads must keep running meanwhile.
foreign import ccall safe "pt_create"       c_pt_create       :: Ptr (Ptr PtCtx) -> Ptr PtParams -> IO CInt
foreign import ccall safe "pt_scene_upload" c_pt_scene_upload :: Ptr PtCtx -> Ptr PtObject -> Int32 -> IO CInt
foreign import ccall safe "pt_trace"        c_pt_trace        :: Ptr PtCtx -> Int32 -> IO CInt
foreign import ccall safe "pt_reset"        c_pt_reset        :: Ptr PtCtx -> IO CInt
foreign import ccall safe "pt_samples_done" c_pt_samples_done :: Ptr PtCtx -> Ptr Int64 -> IO CInt
foreign import ccall safe "pt_scene_specialise" c_pt_scene_specialise :: Ptr PtCtx -> IO CInt
foreign import ccall safe "pt_read_sum"     c_pt_read_sum     :: Ptr PtCtx -> Ptr Double -> IO CInt
foreign import ccall safe "pt_last_error"   c_pt_last_error   :: Ptr PtCtx -> IO CString
foreign import ccall safe "&pt_destroy"     p_pt_destroy      :: FunPtr (Ptr PtCtx -> IO ())

check :: Ptr PtCtx -> CInt -> IO ()
check _ 0 = return ()
check ctx st = do
  msg <- c_pt_last_error ctx >>= peekCString
  ioError (userError ("libpt status " ++ show st ++ ": " ++ msg))

-- One context for the process, created on first use from Params.* and Object.objects.
-- (Several GPUs: call pt_create_multi with a device list here instead of pt_create; nothing else changes --
-- include/pt.h, "Multi-GPU".)
{-# NOINLINE context #-}
context :: ForeignPtr PtCtx
context = unsafePerformIO $ alloca $ \pp -> with params $ \ppar -> do
  c_pt_create pp ppar >>= check nullPtr
  ctx <- peek pp
  withArrayLen (map PtObject (V.toList objects)) $ \n arr ->
    c_pt_scene_upload ctx arr (fromIntegral n) >>= check ctx
  -- Object.objects is a constant of the program (src/Object.hs:147-148): let the library compile its kernel
  -- for it.  Optional -- any status but 0 leaves the general kernel in use -- so the result is not checked.
  _ <- c_pt_scene_specialise ctx
  newForeignPtr p_pt_destroy ctx
  where
    params = PtParams (fromIntegral Params.height) (fromIntegral Params.width) Params.cameraDepth
                      (fromIntegral Params.maxBounces) 0 0 0 1 0

canvasLen :: Int
canvasLen = Params.height * Params.width * 3

-- ONE host buffer for the whole process.  `traces` is a CAF and app/Web.hs:43 walks it for as long as the
-- server runs, so an element must not own memory: a buffer per element would grow by H*W*3 doubles per frame
-- served (6 MB at 500x500), for ever.
{-# NOINLINE buffer #-}
buffer :: ForeignPtr Double
buffer = unsafePerformIO (mallocForeignPtrArray canvasLen)

-- Which element of `traces` the buffer holds (-1: none, or being refilled), and a stamp that changes whenever
-- the buffer is about to change and again when it has changed: readers that find the same stamp before and
-- after a read know the buffer was not touched in between.
{-# NOINLINE held #-}
held :: IORef (Int, Int)
held = unsafePerformIO (newIORef (-1, 0))

-- Serialises everything that talks to the library or refills the buffer (a pt_ctx is not thread-safe).
{-# NOINLINE lock #-}
lock :: MVar ()
lock = unsafePerformIO (newMVar ())

-- Under `lock`: make the buffer hold element k.  Rendered on demand: forwards with the pt_trace calls not made
-- yet (so the consecutive elements app/Web.hs pulls cost one traceCanvas each, and `traces !! numIters` of
-- app/Write.hs:28 costs ONE pt_trace numIters and ONE read), backwards -- an element older than the buffer's is
-- asked for again -- with pt_reset and k fresh samples.  The random stream is keyed by (pixel, sample index,
-- bounce), so element k holds the same numbers whenever it is rendered (up to fp32 summation order; bit for bit
-- with PT_FLAG_NO_STEAL): the list stays a value.
fetch :: Int -> IO ()
fetch k = do
  (have, stamp) <- readIORef held
  when (have /= k) $ withForeignPtr context $ \ctx -> do
    writeIORef held (-1, stamp + 1)
    done <- alloca $ \pd -> do
      c_pt_samples_done ctx pd >>= check ctx
      fromIntegral <$> (peek pd :: IO Int64)
    from <- if k < done then c_pt_reset ctx >>= check ctx >> return 0 else return done
    when (k > from) $ c_pt_trace ctx (fromIntegral (k - from)) >>= check ctx
    withForeignPtr buffer $ \buf -> c_pt_read_sum ctx buf >>= check ctx
    writeIORef held (k, stamp + 2)

-- Value i of element k, the slow way: under the lock, fetch and read (so no other thread can refill the buffer
-- between the two).
readLocked :: Int -> Int -> IO Double
readLocked k i = withMVar lock $ \_ -> do
  fetch k
  withForeignPtr buffer $ \buf -> peekElemOff buf i

-- Value i of element k.  While the buffer holds element k -- every index of app/Write.hs's `traces !! numIters`
-- and of each frame app/Web.hs serves, but the first -- a read is two IORef reads around one peek and takes no
-- lock: 6.2 M indices per 1080p element would otherwise each pay an MVar round trip.  A stamp that moved means
-- another thread refilled the buffer meanwhile (two elements read at the same time); the slow path then gives
-- the right value.  Elements read INTERLEAVED (R.zipWith over traces !! a and traces !! b) are correct but
-- re-rendered at every switch -- copy one first (R.computeS).
readAt :: Int -> Int -> IO Double
readAt k i = do
  (h1, s1) <- readIORef held
  if h1 /= k
    then readLocked k i
    else do
      x <- withForeignPtr buffer $ \buf -> peekElemOff buf i
      (_, s2) <- readIORef held
      if s1 == s2 then return x else readLocked k i

-- Element k of `traces`: a delayed array that holds nothing but k.
element :: Int -> Array D DIM3 Double
element k = R.fromFunction sh $ \ix -> unsafeDupablePerformIO (readAt k (R.toIndex sh ix))
  where
    sh = Z :. Params.height :. Params.width :. 3

-- traces :: [Array D DIM3 Double]  (src/Lib.hs:53-56): element k is the SUM of
-- k samples per pixel; element 0 is the black canvas.
-- (For the byte image, pt_read_rgb8 / pt_read_rgb8_noalloc with PT_RGB8_REFERENCE do repa3ToImage's work on
-- the GPU -- Conversion.hs:22-28 -- and skip the per-index reads above; INTEGRATION.md.)
traces :: [Array D DIM3 Double]
traces = map element [0 ..]
